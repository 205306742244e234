"""The image chain alone (stage entry ptx_img_proxy, bf16 features of the benchmark shape) in a loop, for a rocprofv3
kernel trace: do the idle gaps around k_img_pool belong to these kernels or to the forward's stream structure?
    rocprofv3 --kernel-trace --output-format csv -d /tmp/p -o k -- python scratch/imgchain_tl.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch
import bench
from proxytransformation_amd.synth import CONFIGS
from gpu_util import Stages

cfg = CONFIGS["cfg2"]
B = 4
mod, _ = bench.build_module(cfg, torch.device("cuda:0"))
st = Stages(mod, B, cfg.N, cfg.L, cfg.V)
g = torch.Generator().manual_seed(0)
hw = cfg.img_spacial_dim ** 2
img = torch.randn(B * cfg.V, cfg.input_dim, hw, generator=g).to(torch.bfloat16).cuda()
st.shape.img_dtype = 1
for _ in range(40):
    out = st.img_proxy(img)
torch.cuda.synchronize()
print(out.float().abs().mean().item())
