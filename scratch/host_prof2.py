import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from bench import build_module
from proxytransformation_amd.synth import CONFIGS, make_scene_batch
import proxytransformation_amd.module as M
cfg = CONFIGS['cfg2']; dev = torch.device('cuda')
mod, sd = build_module(cfg, dev)
pts, text, mask, img = make_scene_batch(cfg)
points = [torch.from_numpy(p).to(dev) for p in pts]
td = {"text_feats": torch.from_numpy(text).to(dev), "text_token_mask": torch.from_numpy(mask).to(dev)}
im = torch.from_numpy(img).to(dev).bfloat16()
with torch.no_grad():
    for _ in range(10): mod(points, td, im)
    torch.cuda.synchronize()
    # time individual host pieces
    def T(f, n=2000):
        t0=time.perf_counter()
        for _ in range(n): f()
        return (time.perf_counter()-t0)/n*1e6
    print("check_inputs", T(lambda: mod._check_inputs(points, td, im)))
    print("weights_key", T(lambda: mod._weights_key()))
    print("current_stream", T(lambda: torch.cuda.current_stream(dev)))
    print("current_device", T(lambda: torch.cuda.current_device()))
    print("empty", T(lambda: torch.empty((4,100000,3), dtype=torch.float32, device=dev)))
    c = mod._counts_host
    print("tolist", T(lambda: c[:4].tolist()))
    out = torch.empty((4,100000,3), device=dev)
    print("views", T(lambda: [out[b, :99000] for b in range(4)]))
    st = torch.cuda.current_stream(dev)
    print("sync(idle)", T(lambda: st.synchronize()))
    t0=time.perf_counter()
    for _ in range(200): mod(points, td, im)
    torch.cuda.synchronize(); print("step us", (time.perf_counter()-t0)/200*1e6)
