"""r05: does the eval forward keep its speed and its memory over a long run?  argv: forwards [config]"""
import gc, sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from proxytransformation_amd import MODELS
from proxytransformation_amd.synth import CONFIGS, fill_state_dict, make_scene_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
cfg = CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cfg2"]
dev = torch.device("cuda:0")
mod = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
mod.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(mod.state_dict()).items()})
mod = mod.to(dev).eval()
pts, text, mask, img = make_scene_batch(cfg)
args = ([torch.from_numpy(p).to(dev) for p in pts],
        {"text_feats": torch.from_numpy(text).to(dev), "text_token_mask": torch.from_numpy(mask).to(dev)},
        torch.from_numpy(img).to(dev).to(torch.bfloat16))
with torch.no_grad():
    for _ in range(10):
        mod(*args)
    torch.cuda.synchronize()
    for blk in range(n // 1000):
        t0 = time.perf_counter()
        for _ in range(1000):
            outs = mod(*args)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ms = torch.cuda.memory_stats()
        print(f"forwards {blk * 1000:6d}+: {(t1 - t0):7.4f} ms/forward  allocated {ms['allocated_bytes.all.current'] >> 20} MiB reserved "
              f"{ms['reserved_bytes.all.current'] >> 20} MiB live blocks {ms['allocation.all.current']}  py objects {len(gc.get_objects())}", flush=True)
mod.check()
