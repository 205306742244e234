"""Host time spent INSIDE the library calls of one training step (ctypes call by call) vs the Python around them"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxytransformation_amd import MODELS, train as T, _abi
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
sec = collections.defaultdict(float)
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
    sec["zero_grad"] += t1 - t0; sec["module forward (total)"] += t2 - t1; sec["autograd.backward (total)"] += t3 - t2
def wrap_static(cls, name, label):
    fn = getattr(cls, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); sec[label] += time.perf_counter() - t0; return r
    setattr(cls, name, staticmethod(w))
wrap_static(T._TrainStep, "forward", "  _TrainStep.forward body")
wrap_static(T._TrainStep, "backward", "  _TrainStep.backward body")
for c_, l_ in ((T._BlockFused, "block"), (T._ImgPool, "imgpool"), (T._SlotNet, "slotnet"), (T._AffineApply, "affine")):
    wrap_static(c_, "forward", f"    {l_} fwd body"); wrap_static(c_, "backward", f"    {l_} bwd body")
for _ in range(10): step()
torch.cuda.synchronize()
sec.clear()
lib = _abi.lib()
tm = collections.defaultdict(float); cnt = collections.Counter()
class Wrap:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); tm[self.name] += time.perf_counter() - t0; cnt[self.name] += 1; return r
class LibProxy:
    def __init__(self, lib): self._lib = lib; self._c = {}
    def __getattr__(self, name):
        if name not in self._c: self._c[name] = Wrap(name, getattr(self._lib, name))
        return self._c[name]
proxy = LibProxy(lib)
_abi._lib = proxy
# torch.empty / zeros time
te = [0.0, 0]
orig_empty = torch.empty
def timed_empty(*a, **k):
    t0 = time.perf_counter(); r = orig_empty(*a, **k); te[0] += time.perf_counter() - t0; te[1] += 1; return r
torch.empty = timed_empty
n = 100
sec.clear()
t_all = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
tot = sum(tm.values())
print(f"step {1e3 * t_all / n:.3f} ms; inside library calls {1e3 * tot / n:.3f} ms ({sum(cnt.values()) // n} calls); torch.empty {1e3 * te[0] / n:.3f} ms ({te[1] // n} calls)")
for k, v in sorted(tm.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {k:28s} {1e6 * v / n:7.1f} us  ({cnt[k] // n}x)")
for k, v in sec.items():
    print(f"{k:34s} {1e6 * v / n:8.1f} us")
