"""cProfile of the host side of the training step with autograd's backward in the calling thread"""
import cProfile, pstats, os, sys, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxytransformation_amd import MODELS
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch
cfg = PreshapeConfig("cfg4train", B=6, N=100000, grid_size=12, dynamic_drop_radio=0.6, L=20, V=20, text_blocks=3, img_blocks=3, seed_base=4500)
m = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
m.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(m.state_dict()).items()})
m = m.cuda().train()
pts, text, mask, img = make_scene_batch(cfg)
dev = torch.device("cuda:0")
args = ([torch.from_numpy(p).to(dev) for p in pts], {"text_feats": torch.from_numpy(text).to(dev).requires_grad_(True),
        "text_token_mask": torch.from_numpy(mask).to(dev)}, torch.from_numpy(img).to(dev).requires_grad_(True))
leaves = list(m.parameters()) + [args[1]["text_feats"], args[2]]
gos = None
def step():
    global gos
    for t in leaves: t.grad = None
    outs = m(*args)
    if gos is None: gos = [torch.ones_like(o) for o in outs]
    torch.autograd.backward(outs, gos)
torch.autograd.set_multithreading_enabled(False)
for _ in range(3): step()
torch.cuda.synchronize()
# segment timers (host only, GPU drained before each)
import proxytransformation_amd.train as T
n = 10
acc = dict(zero=0.0, fwd=0.0, loss=0.0, bwd=0.0)
for _ in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in leaves: t.grad = None
    t1 = time.perf_counter()
    outs = m(*args)
    t2 = time.perf_counter()
    t3 = time.perf_counter()
    torch.autograd.backward(outs, gos)
    t4 = time.perf_counter()
    acc["zero"] += t1 - t0; acc["fwd"] += t2 - t1; acc["loss"] += t3 - t2; acc["bwd"] += t4 - t3
print({k: round(1e3 * v / n, 3) for k, v in acc.items()}, "ms per step (host, GPU idle at entry)")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:9000])
