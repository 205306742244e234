"""r05 experiment, second form: the chunk loop enqueued from C++ (needs the lab chunk loop in ptx_img_proxy that commit "r05 chunk overlap lab" carried: see profiles/r05_img_chunk_overlap.txt)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxytransformation_amd.synth import CONFIGS, PreshapeConfig
from tests.util import build_module
from tests.gpu_util import Stages
base = CONFIGS["cfg2"]
TOTAL = int(sys.argv[1])
dev = torch.device("cuda:0")
sets = [torch.randn(TOTAL, base.V, 512, 15, 15, device=dev).to(torch.bfloat16) for _ in range(3)]
cfg = PreshapeConfig("c", B=TOTAL, N=base.N, grid_size=8, dynamic_drop_radio=0.5, L=64, V=base.V)
m, _ = build_module(cfg); m = m.cuda()
sg = Stages(m, TOTAL, cfg.N, cfg.L, cfg.V); sg.shape.img_dtype = 1
for i in range(3): sg.img_proxy(sets[i % 3])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(12): sg.img_proxy(sets[i % 3])
torch.cuda.synchronize()
print(f"{TOTAL} scenes chunk {os.environ.get('PTX_LAB_IMG_CHUNK','-')} streams {os.environ.get('PTX_LAB_IMG_STREAMS','1')}: {1e6*(time.perf_counter()-t0)/12:8.1f} us")
