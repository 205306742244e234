"""r06 (VERDICT r05 "next" #5): k_cluster / k_minmax / k_select where the ball queries cannot stop early -- large extents and a
"two-blob" cloud -- next to the benchmark distribution.  Per case: kernel time inside the forward (events on the kernel's own packet,
ptx_timing_*), the largest scanned index P_max and the share of centres that never fill, and k_cluster's bytes against 12 N per pass.
usage: python scratch/cluster_regimes.py [scenes]"""
import ctypes, dataclasses, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from proxytransformation_amd import _abi
from proxytransformation_amd.synth import CONFIGS, make_scene_batch
from tests.util import build_module

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = _abi.lib()
names = [lib.ptx_kernel_name(i).decode() for i in range(lib.ptx_kernel_count())]
dev = torch.device("cuda:0")
cases = []
for base in ("cfg2", "cfg4"):
    c = CONFIGS[base]
    cases += [(base + " uniform (12,12,9)", c),
              (base + " uniform (40,40,12)", dataclasses.replace(c, extent=(40.0, 40.0, 12.0))),
              (base + " two-blob (30,30,30)", dataclasses.replace(c, extent=(30.0, 30.0, 30.0), distribution="two_blob"))]
print(f"{B} scenes per call; us per launch inside the forward")
for name, cfg in cases:
    V = min(cfg.V, 8)                                            # the image side is not what is measured here
    cfg = dataclasses.replace(cfg, B=B, V=V)
    m, _ = build_module(cfg)
    m = m.cuda()
    pts, text, mask, img = make_scene_batch(cfg)
    t = lambda a: torch.from_numpy(a).to(dev)
    args = ([t(p) for p in pts], {"text_feats": t(text), "text_token_mask": t(mask)}, t(img).to(torch.bfloat16))
    with torch.no_grad():
        for _ in range(3):
            m(*args)
        d = m.forward_debug(*args)
        idx2 = d["idx2"].cpu().numpy()
        pmax = int(idx2.max())
        never = float((idx2[:, :, -1] < 0).mean())
        nk = len(names)
        lib.ptx_timing_every(1)
        lib.ptx_timing_select_mask((1 << nk) - 1)
        for _ in range(10):
            m(*args)
        torch.cuda.synchronize()
        n = (ctypes.c_int * nk)(); ms = (ctypes.c_float * nk)()
        lib.ptx_timing_read_sites(n, ms, nk)
        lib.ptx_timing_select(-1)
    us = {names[i]: 1e3 * ms[i] / n[i] for i in range(nk) if n[i] > 0}
    kc = us.get("k_cluster", float("nan"))
    full = 2 * 12 * cfg.N * cfg.B                                  # both query passes read every point once
    print(f"{name:28s} k_minmax {us.get('k_minmax', float('nan')):7.1f}  k_cluster {kc:8.1f}  k_select {us.get('k_select', float('nan')):7.1f}   "
          f"P_max {pmax:6d}  centres never full {never:5.2f}   k_cluster vs 2 x 12 N: {full / kc / 1e3:7.1f} GB/s "
          f"({full / kc / 1e3 / 8000:.4f} of HBM)", flush=True)
