"""Host time of the sections of the training forward and of the backward nodes (perf_counter, no device sync inside a step)"""
import os, sys, time, collections
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
gos = None
torch.autograd.set_multithreading_enabled(False)
# wrap the backward of every node class of train.py
bw = collections.defaultdict(float); bwn = collections.Counter()
for name in dir(T):
    cls = getattr(T, name)
    if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
        orig = cls.backward
        def make(orig, name):
            def wrapped(ctx, *a):
                t0 = time.perf_counter(); r = orig(ctx, *a); bw[name] += time.perf_counter() - t0; bwn[name] += 1; return r
            return staticmethod(wrapped)
        cls.backward = make(orig, name)
def step(rec=None):
    global gos
    t0 = time.perf_counter()
    for t in leaves: t.grad = None
    t1 = time.perf_counter()
    outs = m(*args)
    t2 = time.perf_counter()
    if gos is None: gos = [torch.ones_like(o) for o in outs]
    torch.autograd.backward(outs, gos)
    t3 = time.perf_counter()
    if rec is not None: rec["zero"] += t1 - t0; rec["forward"] += t2 - t1; rec["backward"] += t3 - t2
for _ in range(3): step()
torch.cuda.synchronize()
bw.clear(); bwn.clear()
n = 20
rec = collections.defaultdict(float); sec = collections.defaultdict(float)
t_all = time.perf_counter()
for _ in range(n):
    T._TICKS = []
    step(rec)
    tk = T._TICKS
    for (a, ta), (b, tb) in zip(tk[:-1], tk[1:]): sec[b] += tb - ta
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
print(f"step {1e3 * t_all / n:.3f} ms;  host: " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in rec.items()))
print("forward sections (us): " + ", ".join(f"{k} {1e6 * v / n:.0f}" for k, v in sec.items()))
print("backward nodes (us per step): " + ", ".join(f"{k} {1e6 * v / n:.0f} ({bwn[k] // n}x)" for k, v in sorted(bw.items(), key=lambda kv: -kv[1])))
