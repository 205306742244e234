"""Host side of the benchmark forward: time to ENQUEUE a forward (the call returns as soon as the survivor counts are
published, i.e. it includes the wait for k_tags) against the GPU time per step: python scratch/host_enqueue.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from proxytransformation_amd.synth import CONFIGS
cfg = CONFIGS["cfg2"]; dev = torch.device("cuda:0")
mod, _ = bench.build_module(cfg, dev)
inp = bench.InputSets(cfg, 4, 3, 0, 1, dev, torch.bfloat16)
with torch.no_grad():
    for i in range(40):
        mod(*inp.args(i))
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for i in range(n):
        mod(*inp.args(i))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"per step: host loop {1e6 * (t1 - t0) / n:.1f} us, with the final drain {1e6 * (t2 - t0) / n:.1f} us")
    # pure host work: the same loop with the GPU kept idle between calls
    tot = 0.0
    for i in range(100):
        torch.cuda.synchronize()
        a = time.perf_counter()
        mod(*inp.args(i))
        tot += time.perf_counter() - a
    print(f"one forward from an idle GPU until the call returns: {1e6 * tot / 100:.1f} us")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(200):
        mod(*inp.args(i))
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
