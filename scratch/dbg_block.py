import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxytransformation_amd import MODELS, train as T
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict
dev = torch.device("cuda:0")
def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g).to(dev)
embed = 256
cfg = PreshapeConfig("fb", B=3, N=2000, grid_size=4, dynamic_drop_radio=0.5, L=6, V=3, embed_dim=embed, seed_base=8600)
res = []
for fused in (True, False):
    T._FUSED_BLOCK = fused
    m = MODELS.build(dict(type="ProxyTransformationNormReverse", drop_rate=0., attn_drop_rate=0., drop_path_rate=0., **cfg.module_kwargs()))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(m.state_dict()).items()})
    m = m.cuda().train()
    B, n, L = 3, m.real_cluster_num, 6
    x = rnd(B * n, embed, seed=51).requires_grad_(True); proxy = rnd(B * L, embed, seed=52).requires_grad_(True)
    mask = torch.ones(B, L, dtype=torch.uint8, device=dev); mask[1, 4:] = 0
    seeds = T.site_seeds(99, 1, 1)[0]
    xa, xb = (x, x) if fused else T.fork(x, 2)
    t = T._block(m, m.imgformer[-1], m.img_norm[-1], m.img_trans, m.img_trans_norm, xa, xb, proxy, mask, B, n, L, seeds)
    t.backward(rnd(B * n, 9, seed=53))
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    grads.update({"x": x.grad, "proxy": proxy.grad})
    res.append(grads)
for k in res[0]:
    a, b = res[0][k].double().cpu().numpy().ravel(), res[1][k].double().cpu().numpy().ravel()
    print(f"{k:40s} err/rms {np.abs(a-b).max()/(np.sqrt((b**2).mean())+1e-30):.3e}  rms {np.sqrt((b**2).mean()):.3e}  a[:3] {a[:3]} b[:3] {b[:3]}")
