"""Timings of the stages next to the path: voxel quantisation (N2) and image->point sampling (N3)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_module, InputSets
from proxytransformation_amd.synth import CONFIGS
from proxytransformation_amd.fusion import batch_point_sample
cfg = CONFIGS["cfg2"]; dev = torch.device("cuda:0")
mod, _ = build_module(cfg, dev)
inp = InputSets(cfg, 4, 1, 0, 1, dev, torch.bfloat16)
with torch.no_grad():
    outs = mod(*inp.args(0))
    for _ in range(3): mod.quantize(outs, 0.01)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c, f = mod.quantize(outs, 0.01)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    npts = sum(o.shape[0] for o in outs)
    print(f"quantize: {npts} points -> {c.shape[0]} voxels in {dt*1e6:.0f} us ({npts*12*2/dt/1e9:.0f} GB/s of xyz in+out, host sync included)")
V, C, H, W, N = 50, 256, 30, 30, 200000
feats = torch.randn(V, C, H, W, device=dev)
proj = torch.eye(4, device=dev).repeat(V, 1, 1); proj[:, 0, 0] = proj[:, 1, 1] = 300; proj[:, 0, 2] = 320; proj[:, 1, 2] = 240; proj[:, 2, 3] = 4.0
pts = (torch.rand(N, 3, device=dev) - 0.5) * 6
for _ in range(3): batch_point_sample(None, feats, pts, proj)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): o = batch_point_sample(None, feats, pts, proj)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"point_sample: N={N} V={V} C={C} {H}x{W}: {dt*1e6:.0f} us ({N*C*4/dt/1e9:.0f} GB/s of output rows)")
