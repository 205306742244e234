"""ptx_voxelize at the benchmark's size (4 scenes x ~100k surviving points, 1 cm voxels): python scratch/vox_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxytransformation_amd.synth import CONFIGS, PreshapeConfig, make_scene_batch
from tests.util import build_module
cfg = CONFIGS["cfg2"]
small = PreshapeConfig("vx", B=4, N=cfg.N, grid_size=cfg.grid_size, dynamic_drop_radio=cfg.dynamic_drop_radio, L=4, V=2, seed_base=cfg.seed_base)
m, _ = build_module(small); m = m.cuda()
pts, text, mask, img = make_scene_batch(small)
t = lambda a: torch.from_numpy(a).cuda()
with torch.no_grad():
    outs = m([t(p) for p in pts], {"text_feats": t(text), "text_token_mask": t(mask)}, t(img))
    for _ in range(3): c, f = m.quantize(outs, 0.01)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps): c, f = m.quantize(outs, 0.01)
    e1.record(); torch.cuda.synchronize()
print(f"quantize: {sum(o.shape[0] for o in outs)} points -> {c.shape[0]} voxels, {1e3 * e0.elapsed_time(e1) / reps:.1f} us per call (stream time, incl. the host's count wait per call)")
