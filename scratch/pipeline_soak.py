"""r06: the chained configs[3] call in a loop at the shipped shape -- allocator state (live bytes / blocks, reserved) after every 50
calls, per-call wall time, and the outputs of the last call against the first (bit for bit).  usage: python scratch/pipeline_soak.py [calls]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from proxytransformation_amd.pipeline import GroundingFeaturePrefix
from proxytransformation_amd.synth import CONFIGS, FPN_LEVELS, make_depth_scene
from tests.util import build_module
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = CONFIGS["cfg4_room"]
dev = torch.device("cuda:0")
m, _ = build_module(cfg)
m = m.cuda()
B, V = 6, cfg.V
scenes = []
for b in range(B):
    sc = make_depth_scene(9100 + b, V=V)
    scenes.append(dict(sc, depth_img=torch.from_numpy(sc["depth_img"].view(np.int16)).to(dev).view(torch.uint16)))
g = torch.Generator(device=dev); g.manual_seed(1)
feats = [torch.randn((B, V, c, s, s), generator=g, device=dev) for c, s in FPN_LEVELS]
text = {"text_feats": torch.randn((B, cfg.L, cfg.embed_dim), generator=g, device=dev), "text_token_mask": torch.ones((B, cfg.L), dtype=torch.bool, device=dev)}
pipe = GroundingFeaturePrefix(m, n_points=cfg.N)
first = pipe(scenes, text, feats, rng=np.random.RandomState(0))
pre = [dict(sc, choices=first.ingested.sel[b]) for b, sc in enumerate(scenes)]
ref = pipe(pre, text, feats)
torch.cuda.synchronize()
def state():
    s = torch.cuda.memory_stats()
    return s["allocated_bytes.all.current"] >> 20, s["allocation.all.current"], s["reserved_bytes.all.current"] >> 20
print("call   ms/call  allocated MiB  blocks  reserved MiB")
t0 = time.perf_counter()
for i in range(1, n_calls + 1):
    res = pipe(pre, text, feats)
    if i % 50 == 0:
        torch.cuda.synchronize()
        print(f"{i:5d} {1e3 * (time.perf_counter() - t0) / 50:8.3f} {state()[0]:12d} {state()[1]:7d} {state()[2]:12d}", flush=True)
        t0 = time.perf_counter()
torch.cuda.synchronize()
m.check()
same = torch.equal(res.coordinates, ref.coordinates) and all(torch.equal(a, b) for ra, rb in zip(res.points_imgfeats, ref.points_imgfeats) for a, b in zip(ra, rb)) \
    and all(torch.equal(a, b) for a, b in zip(res.points, ref.points))
print("last call == first call, bit for bit:", same)
