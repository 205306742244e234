"""cProfile of the training step's host side (relative weights only: the profiler slows everything)"""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxytransformation_amd import MODELS
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch
cfg = PreshapeConfig("cfg4train", B=6, N=100000, grid_size=12, dynamic_drop_radio=0.6, L=20, V=20, text_blocks=3, img_blocks=3, seed_base=4500)
m = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
m.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(m.state_dict()).items()})
m = m.cuda().train()
pts, text, mask, img = make_scene_batch(cfg)
dev = torch.device("cuda:0")
args = ([torch.from_numpy(p).to(dev) for p in pts], {"text_feats": torch.from_numpy(text).to(dev).requires_grad_(True),
        "text_token_mask": torch.from_numpy(mask).to(dev)}, torch.from_numpy(img).to(dev).requires_grad_(True))
leaves = list(m.parameters()) + [args[1]["text_feats"], args[2]]
torch.autograd.set_multithreading_enabled(False)
gos = None
def step():
    global gos
    for t in leaves: t.grad = None
    outs = m(*args)
    if gos is None: gos = [torch.ones_like(o) for o in outs]
    torch.autograd.backward(outs, gos)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
