"""Host cost of one forward: the ptx_forward enqueue alone, the python wrapper, the count wait; 1 and 2 streams."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxytransformation_amd import _abi
from bench import build_module, InputSets
from proxytransformation_amd.synth import CONFIGS
cfg = CONFIGS["cfg2"]
dev = torch.device("cuda:0")
mod, sd = build_module(cfg, dev)
inp = InputSets(cfg, 4, 3, 0, 1, dev, torch.bfloat16)
lib = _abi.lib()
orig_fwd, orig_wait = lib.ptx_forward, lib.ptx_wait_counts
acc = {"fwd": 0.0, "wait": 0.0}
class Wrap:
    def __init__(self, f, k): self.f, self.k = f, k
    def __call__(self, *a):
        t = time.perf_counter(); r = self.f(*a); acc[self.k] += time.perf_counter() - t; return r
lib.ptx_forward = Wrap(orig_fwd, "fwd"); lib.ptx_wait_counts = Wrap(orig_wait, "wait")
n = int(os.environ.get("N", 200))
for ns in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for i in range(12):
        with torch.cuda.stream(streams[i % ns]):
            mod(*inp.args(i))
    torch.cuda.synchronize()
    acc["fwd"] = acc["wait"] = 0.0
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i % ns]):
            mod(*inp.args(i))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{ns} stream(s): per call host {1e6*(t1-t0)/n:.1f} us (+ drain {1e6*(t2-t1):.0f} us once); ptx_forward {1e6*acc['fwd']/n:.1f}; "
          f"wait_counts {1e6*acc['wait']/n:.1f}; python rest {1e6*((t1-t0)-acc['fwd']-acc['wait'])/n:.1f};  => {4*n/(t2-t0):.0f} scenes/s")
