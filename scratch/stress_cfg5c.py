"""cfg5's clustering shape (500k points, 16^3 grid -> 1024 kept clusters, 1844 FPS picks) at the supported width
(d = 256): does the path hold up and where does the time go?  python scratch/stress_cfg5c.py"""
import ctypes, json, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxytransformation_amd import MODELS, _abi
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch

cfg = PreshapeConfig("cfg5c", B=1, N=500000, grid_size=16, dynamic_drop_radio=0.75, L=64, V=192, seed_base=5000)
mod = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
sd = fill_state_dict(mod.state_dict())
mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
mod = mod.eval().cuda()
pts, text, mask, img = make_scene_batch(cfg)
dev = torch.device("cuda")
args = ([torch.from_numpy(p).to(dev) for p in pts], {"text_feats": torch.from_numpy(text).to(dev),
        "text_token_mask": torch.from_numpy(mask).to(dev)}, torch.from_numpy(img).to(dev).to(torch.bfloat16))
with torch.no_grad():
    for _ in range(3):
        out = mod(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        out = mod(*args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    lib = _abi.lib(); bd = {}
    for i in range(lib.ptx_kernel_count()):
        lib.ptx_timing_select(i)
        for _ in range(3):
            mod(*args)
        torch.cuda.synchronize()
        n, ms = ctypes.c_int(0), ctypes.c_float(0)
        lib.ptx_timing_read(ctypes.byref(n), ctypes.byref(ms))
        bd[lib.ptx_kernel_name(i).decode()] = round(1e3 * ms.value / max(n.value, 1), 1)
    lib.ptx_timing_select(-1)
print(json.dumps(dict(ms_per_forward=round(dt * 1e3, 3), kept=int(out[0].shape[0]), breakdown_us=bd)))
