"""r05: run-to-run determinism of the training step's two forms (one autograd node / per-operator graph) under allocator churn, with
a loss that reaches the transforms from two sources.  Any bitwise difference between runs of the same form is a stream race."""
import copy, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proxytransformation_amd.train as T
from proxytransformation_amd import MODELS
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch
from tests.gpu_util import t
from tests.test_gpu_train import _loss
cfg = PreshapeConfig("trx", B=3, N=5000, grid_size=5, dynamic_drop_radio=0.6, L=9, V=4, seed_base=8900)
pts, text, mask, img = make_scene_batch(cfg)
m0 = MODELS.build(dict(type="ProxyTransformationNormReverse", drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, **cfg.module_kwargs()))
m0.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(m0.state_dict()).items()})
g = torch.Generator().manual_seed(5)
dev = torch.device("cuda:0")
wk, wt, wf = (torch.randn(cfg.B, cfg.M_keep, n, generator=g).to(dev) for n in (3, 3, 9))
rnd = random.Random(1)
def churn():
    junk = [torch.full((rnd.choice([1 << 10, 1 << 14, 1 << 18, 300000, 70000, 1 << 20]),), float("nan"), device=dev) for _ in range(rnd.randint(0, 6))]
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        junk.append(torch.full((rnd.choice([1 << 12, 1 << 16, 1 << 19]),), float("nan"), device=dev))
    del junk
def run(one_node, img_grad):
    T._ONE_NODE = one_node
    m = copy.deepcopy(m0).cuda().train()
    churn()
    tx = t(text).requires_grad_(True)
    ix = t(img).requires_grad_(img_grad)
    outs, tf = m([t(p) for p in pts], {"text_feats": tx, "text_token_mask": t(mask)}, ix, return_transforms=True)
    loss = (tf["kcenter"] * wk).sum() + (tf["translate"] * wt).sum() + (tf["transform"] * wf).sum() + _loss(outs)
    churn()
    loss.backward()
    churn()
    res = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    res["tx"] = tx.grad.clone()
    if img_grad: res["ix"] = ix.grad.clone()
    torch.cuda.synchronize()
    return res
for one in (True, False):
    for ig in (False, True):
        base = run(one, ig)
        bad = {}
        for it in range(40):
            r = run(one, ig)
            for k in base:
                if not torch.equal(r[k], base[k]):
                    bad[k] = bad.get(k, 0) + 1
        print(f"one_node={one} img_grad={ig}: 40 repeats, tensors that differed from the first run: {bad if bad else 'none'}", flush=True)
# the two stream arrangements of the one-node step give the same bits
T._BLOCKS_APART = True
a = run(True, True)
T._BLOCKS_APART = False
b = run(True, True)
T._BLOCKS_APART = True
diff = [k for k in a if not torch.equal(a[k], b[k])]
print("blocks apart vs serial, tensors that differ:", diff if diff else "none")
