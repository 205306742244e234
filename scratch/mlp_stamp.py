"""Phase stamps of the four slices of one row tile of k_mlp inside the bench forward, from a lab build exporting ptx_lab_mdbg
(stamps are added to a copy of csrc/mlp.hip, never to the product): python scratch/mlp_stamp.py"""
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
buf = np.zeros((4, 8, 12), np.uint64)
raw.ptx_lab_mdbg(ctypes.c_void_p(buf.ctypes.data))
names = ["start", "requests issued", "x1 split + sync", "phase 1 done", "GELU + H planes + sync", "phase 2 done", "partials out + sync",
         "ticket", "merge + heads done"]
for z in range(4):
    t0 = buf[z, :, 0].min()
    print("slice", z)
    for k, n in enumerate(names):
        print(f"{n:>24s} " + " ".join(f"{int(buf[z, w, k]) - int(t0) if buf[z, w, k] else -1:7d}" for w in range(8)))
