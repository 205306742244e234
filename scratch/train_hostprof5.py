"""host time of module.forward around train.forward_train, and of _TrainStep.forward / backward (single-thread autograd)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxytransformation_amd import MODELS, train as T
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
torch.autograd.set_multithreading_enabled(False)
acc = {}
def wrap(obj, name, key):
    orig = getattr(obj, name)
    def f(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, staticmethod(f) if isinstance(obj, type) else f)
wrap(T, "forward_train", "forward_train")
wrap(T._TrainStep, "forward", "TrainStep.forward")
wrap(T._TrainStep, "backward", "TrainStep.backward")
gos = None
def step():
    global gos
    t0 = time.perf_counter()
    for t in leaves: t.grad = None
    t1 = time.perf_counter()
    outs = m(*args)
    t2 = time.perf_counter()
    if gos is None: gos = [torch.ones_like(o) for o in outs]
    torch.autograd.backward(outs, gos)
    t3 = time.perf_counter()
    for k, v in (("zero", t1 - t0), ("module.forward", t2 - t1), ("autograd.backward", t3 - t2)): acc[k] = acc.get(k, 0.0) + v
for _ in range(5): step()
torch.cuda.synchronize(); acc.clear()
n = 40; t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"step {1e3 * tot / n:.3f} ms; " + ", ".join(f"{k} {1e6 * v / n:.0f} us" for k, v in acc.items()))
