"""Phase stamps of two work-groups of the fc1 GEMM (k_gemm64x<2,1,3> with the GELU epilogue) inside the bench forward, from a
lab build exporting ptx_lab_xdbg: python scratch/g64_stamp.py"""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from proxytransformation_amd import _abi
from proxytransformation_amd.synth import CONFIGS
cfg = CONFIGS["cfg2"]; dev = torch.device("cuda:0")
mod, _ = bench.build_module(cfg, dev)
inp = bench.InputSets(cfg, 4, 3, 0, 1, dev, torch.bfloat16)
with torch.no_grad():
    for i in range(12):
        mod(*inp.args(i))
torch.cuda.synchronize()
raw = ctypes.CDLL(_abi.LIB_PATH)
buf = np.zeros((2, 4, 8), np.uint64)
raw.ptx_lab_xdbg(ctypes.c_void_p(buf.ctypes.data))
names = ["start", "LN statistics", "fetches issued", "first stash + sync", "K loop done", "epilogue done"]
for g in range(2):
    t0 = buf[g, :, 0].min()
    print("work-group", g)
    for k, n in enumerate(names):
        print(f"{n:>20s} " + " ".join(f"{int(buf[g, w, k]) - int(t0) if buf[g, w, k] else -1:7d}" for w in range(4)))
