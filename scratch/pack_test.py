import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from bench import build_module
from proxytransformation_amd.synth import CONFIGS, make_scene_batch
cfg = CONFIGS['cfg2']; dev = torch.device('cuda')
pts, text, mask, img = make_scene_batch(cfg)
points = [torch.from_numpy(p).to(dev) for p in pts]
td = {"text_feats": torch.from_numpy(text).to(dev), "text_token_mask": torch.from_numpy(mask).to(dev)}
im = torch.from_numpy(img).to(dev)
def run(pack):
    mod, sd = build_module(cfg, dev)
    if pack:
        ts = [t for t in mod.state_dict(keep_vars=True).values() if t.is_floating_point()]
        tot = sum((t.numel()+63)//64*64 for t in ts)
        flat = torch.empty(tot, device=dev)
        off = 0
        for t in ts:
            n = t.numel()
            flat[off:off+n].copy_(t.detach().reshape(-1))
            t.data = flat[off:off+n].view(t.shape)
            off += (n+63)//64*64
        mod._tensors = None; mod._wkey = None
    with torch.no_grad():
        for _ in range(10): mod(points, td, im)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(100): mod(points, td, im)
        torch.cuda.synchronize(); t1=time.perf_counter()
    print("pack" if pack else "plain", (t1-t0)/100*1e6, "us/step")
run(False); run(True); run(False); run(True)
