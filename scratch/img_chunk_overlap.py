"""r05 experiment: at many scenes per call, does the image chain of chunk i+1 (mean pass: HBM-bound) overlap
with the pooling pass of chunk i (request-bound; chunk small enough to come out of the Infinity Cache)?  Stage entry point
ptx_img_proxy on 1 / 2 / 3 torch streams, chunks of 2 / 4 / 8 scenes out of 32, against the whole batch in one call."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxytransformation_amd.synth import CONFIGS, PreshapeConfig
from tests.util import build_module
from tests.gpu_util import Stages

base = CONFIGS["cfg2"]
TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
# three rotating input sets so that nothing of the previous iteration is cache resident
sets = [torch.randn(TOTAL, base.V, 512, 15, 15, device=dev).to(torch.bfloat16) for _ in range(3)]

def mk(chunk):
    cfg = PreshapeConfig("c", B=chunk, N=base.N, grid_size=8, dynamic_drop_radio=0.5, L=64, V=base.V)
    m, _ = build_module(cfg)
    return m.cuda(), cfg

def run(chunk, nstreams, iters=12):
    m, cfg = mk(chunk)
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 0 else [torch.cuda.current_stream()]
    st = []
    for s in streams:
        with torch.cuda.stream(s):
            sg = Stages(m, chunk, cfg.N, cfg.L, cfg.V)
            sg.shape.img_dtype = 1          # bf16-stored features
            st.append(sg)
    nch = TOTAL // chunk
    def once(img):
        ev = torch.cuda.Event(); ev.record()
        for s in streams: s.wait_event(ev)
        for c in range(nch):
            j = c % len(streams)
            with torch.cuda.stream(streams[j]):
                st[j].stream = streams[j].cuda_stream
                st[j].img_proxy(img[c * chunk:(c + 1) * chunk])
        for s in streams: torch.cuda.current_stream().wait_stream(s)
    for i in range(3): once(sets[i % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters): once(sets[i % 3])
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / iters

print(f"{TOTAL} scenes of cfg2 image features (bf16), image chain only (ptx_img_proxy), us per batch")
for chunk, ns in [(TOTAL, 0), (8, 1), (8, 2), (4, 1), (4, 2), (4, 3), (2, 1), (2, 2), (2, 3), (2, 4)]:
    if chunk > TOTAL: continue
    print(f"chunk {chunk:2d} scenes, {max(ns,1)} stream(s): {run(chunk, ns):8.1f} us", flush=True)
