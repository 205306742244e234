import sys, os, cProfile, pstats, time, torch
sys.path.insert(0, os.getcwd())
from bench import build_module
from proxytransformation_amd.synth import CONFIGS, make_scene_batch
cfg = CONFIGS['cfg2']; dev = torch.device('cuda')
mod, sd = build_module(cfg, dev)
pts, text, mask, img = make_scene_batch(cfg)
points = [torch.from_numpy(p).to(dev) for p in pts]
td = {"text_feats": torch.from_numpy(text).to(dev), "text_token_mask": torch.from_numpy(mask).to(dev)}
im = torch.from_numpy(img).to(dev)
with torch.no_grad():
    for _ in range(10): mod(points, td, im)
    torch.cuda.synchronize()
    # host time of one forward when the GPU is idle (no queueing): wall per step and time until launch returns
    t0=time.perf_counter()
    for _ in range(100): mod(points, td, im)
    torch.cuda.synchronize(); t1=time.perf_counter()
    print("wall per step us", (t1-t0)/100*1e6)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): mod(points, td, im)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
