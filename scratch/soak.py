"""Determinism soak: many forwards over rotating input sets, outputs compared bitwise with the first pass (races between
the library's streams would show up as differences)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_module, InputSets
from proxytransformation_amd.synth import CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = CONFIGS[name]; dev = torch.device("cuda:0")
mod, _ = build_module(cfg, dev)
inp = InputSets(cfg, cfg.B, 3, 0, 1, dev, torch.bfloat16)
ref = {}
bad = 0
with torch.no_grad():
    for i in range(n):
        outs = mod(*inp.args(i))
        key = i % 3
        if key not in ref:
            ref[key] = [o.clone() for o in outs]
        else:
            for a, b in zip(outs, ref[key]):
                if a.shape != b.shape or not torch.equal(a, b):
                    bad += 1
                    break
torch.cuda.synchronize()
print(name, "forwards", n, "mismatching", bad)
