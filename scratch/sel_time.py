import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.getcwd())
from proxytransformation_amd.synth import CONFIGS, make_scene_batch, PreshapeConfig
from tests.util import build_module
from tests.gpu_util import Stages, t
def run(ddr):
    cfg = PreshapeConfig("x", B=4, N=100000, grid_size=8, dynamic_drop_radio=ddr, L=64, V=2)
    m,_ = build_module(cfg); m = m.cuda()
    pts = make_scene_batch(cfg)[0]
    st = Stages(m, cfg.B, cfg.N, cfg.L, cfg.V)
    P = t(pts)
    mm, c = st.grid_centers(P)
    idx, cl, pc = st.ball_query(c, P)
    for _ in range(3): o = st.select(idx, c, cl, pc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): o = st.select(idx, c, cl, pc)
    e1.record(); torch.cuda.synchronize()
    print(f"ddr={ddr} Mt={cfg.Mt} Mk={cfg.M_keep} Kd={cfg.Kd}: {e0.elapsed_time(e1)/20*1e3:.1f} us/call (incl. host alloc)")
for ddr in (0.5, 0.3, 0.302):
    run(ddr)
