"""Phase stamps of every unit of k_img_pool from scratch/lab/lib_poolstamp.so (scratch/pool_lab_build.py): where a launch's
time goes -- per-unit phases, residency and phase overlap of the work-groups that share a CU, gaps between units
(profiles/r03_pool_phase_stamps.txt).     python scratch/pool_stamp.py [scenes]"""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import bench
from proxytransformation_amd.synth import CONFIGS
from proxytransformation_amd import _abi
from gpu_util import Stages

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = CONFIGS["cfg2"]
mod, _ = bench.build_module(cfg, torch.device("cuda:0"))
st = Stages(mod, B, cfg.N, cfg.L, cfg.V)
g = torch.Generator().manual_seed(0)
hw = cfg.img_spacial_dim ** 2
imgs = [torch.randn(B * cfg.V, cfg.input_dim, hw, generator=g).to(torch.bfloat16).cuda() for _ in range(3)]
st.shape.img_dtype = 1
for k in range(12):
    st.img_proxy(imgs[k % 3])
torch.cuda.synchronize()
nimg = B * cfg.V
nvb = (nimg + 7) // 8 * 16
dbg = torch.zeros(nvb * 16, dtype=torch.int64, device="cuda")
raw = ctypes.CDLL(_abi.LIB_PATH)
assert raw.ptx_lab_pool_dbg(ctypes.c_void_p(dbg.data_ptr())) == 0
torch.cuda.synchronize()
st.img_proxy(imgs[0])
torch.cuda.synchronize()
raw.ptx_lab_pool_dbg(ctypes.c_void_p(0))
d = dbg.cpu().numpy().reshape(nvb, 16).astype(np.int64)
d = d[d[:, 0] != 0]
NS = 6
names = ["prologue (weights in LDS; the 16 tile requests are issued in it)", "tile wait + scores", "softmax", "weighted sums", "write-out"]
t = d[:, :NS].astype(np.float64) * 0.01          # us
d = np.concatenate([d[:, :6], d[:, 12:14]], axis=1)
LAST = NS - 1; LOADED = 2
t0 = t[:, 0].min()
t -= t0
print(f"scenes {B}: {len(d)} units, "
      f"{len(np.unique(d[:, 7]))} work-groups, span {t[:, LAST].max():.1f} us")
for k, nm in enumerate(names):
    x = t[:, k + 1] - t[:, k]
    print(f"  {nm:66s} mean {x.mean():6.2f}  p10 {np.percentile(x, 10):6.2f}  p90 {np.percentile(x, 90):6.2f} us")
life = t[:, LAST] - t[:, 0]
print(f"  {'unit life':66s} mean {life.mean():6.2f}  p10 {np.percentile(life, 10):6.2f}  p90 {np.percentile(life, 90):6.2f} us;  sum/512 slots = {life.sum() / 512:.1f} us")
# start-time histogram: how the launch fills
st0 = np.sort(t[:, 0])
print("  unit starts by time (us): " + " ".join(f"{np.searchsorted(st0, x):5d}" for x in np.arange(0, t[:, LAST].max() + 5, 5.0)) + "   (cumulative, every 5 us)")
# per CU: residency and phase overlap
cu = (d[:, 6] >> 32) * 65536 + ((d[:, 6] & 0xffffffff) >> 8 & 0xff)
cus = np.unique(cu)
span = t[:, LAST].max()
grid = np.arange(0, span, 0.05)
res = np.zeros((len(cus), len(grid)), np.int8); lod = np.zeros_like(res)
for ci, c in enumerate(cus):
    for r in t[cu == c]:
        res[ci, (grid >= r[0]) & (grid < r[LAST])] += 1
        lod[ci, (grid >= r[0]) & (grid < r[LOADED])] += 1
print(f"  {len(cus)} CUs seen; units per CU min {min((cu == c).sum() for c in cus)} max {max((cu == c).sum() for c in cus)}")
for k in range(4):
    print(f"  CU-time with {k} resident work-groups: {100.0 * (res == k).mean():5.1f} %")
both = (res >= 2)
print(f"  of the time with two residents: both waiting for their tile {100.0 * ((lod >= 2) & both).sum() / max(both.sum(), 1):5.1f} %, "
      f"one waiting {100.0 * ((lod == 1) & both).sum() / max(both.sum(), 1):5.1f} %, none {100.0 * ((lod == 0) & both).sum() / max(both.sum(), 1):5.1f} %")
chip = lod.sum(0)
print("  work-groups waiting for a tile, chip-wide, every 2.5 us: " + " ".join(f"{chip[int(x / 0.05)]:4d}" for x in np.arange(0, span - 0.05, 2.5)))
# gaps: on a CU, time from a unit's end to the next unit start after it (fresh form: dispatch; persistent: none)
gaps = []
for c in cus:
    r = t[cu == c]
    ends = np.sort(r[:, LAST]); starts = np.sort(r[:, 0])
    for e in ends:
        j = np.searchsorted(starts, e - 1e-9)
        if j < len(starts): gaps.append(starts[j] - e)
gaps = np.array(gaps)
print(f"  end of a unit -> next start on the same CU: mean {gaps.mean():5.2f} p50 {np.percentile(gaps, 50):5.2f} p90 {np.percentile(gaps, 90):5.2f} us")
