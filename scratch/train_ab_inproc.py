"""In-process interleaved A/B of a module-level switch of train.py: python scratch/train_ab_inproc.py _BLOCK_STREAMS 0 1
(20-step blocks alternating between the two values; min / median per value -- box phases hit both alike)"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxytransformation_amd import MODELS, train as T
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch
name, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
conv = lambda v: (v != "0") if isinstance(getattr(T, name), bool) else int(v)
cfg = PreshapeConfig("cfg4train", B=6, N=100000, grid_size=12, dynamic_drop_radio=0.6, L=20, V=20, text_blocks=3, img_blocks=3, seed_base=4500)
m = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
m.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(m.state_dict()).items()})
m = m.cuda().train()
pts, text, mask, img = make_scene_batch(cfg)
dev = torch.device("cuda:0")
args = ([torch.from_numpy(p).to(dev) for p in pts], {"text_feats": torch.from_numpy(text).to(dev).requires_grad_(True),
        "text_token_mask": torch.from_numpy(mask).to(dev)}, torch.from_numpy(img).to(dev).requires_grad_(True))
leaves = list(m.parameters()) + [args[1]["text_feats"], args[2]]
gos = {}
def step():
    for t in leaves: t.grad = None
    outs = m(*args)
    key = tuple(o.shape[0] for o in outs)
    if key not in gos: gos[key] = [torch.ones_like(o) for o in outs]
    torch.autograd.backward(outs, gos[key])
res = {va: [], vb: []}
for v in (va, vb):
    setattr(T, name, conv(v))
    for _ in range(5): step()
for rnd in range(int(os.environ.get("ROUNDS", "12"))):
    for v in (va, vb):
        setattr(T, name, conv(v))
        step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t0) / 20)
for v in (va, vb):
    r = res[v]
    print(f"{name}={v}: min {min(r):.3f}  median {statistics.median(r):.3f}  max {max(r):.3f} ms  ({len(r)} blocks of 20 steps)")
