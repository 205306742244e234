"""r05: eager forward vs HIP-graph replay of forward_padded under the runtime's graph knobs (set in the environment by the caller)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from proxytransformation_amd import MODELS
from proxytransformation_amd.synth import CONFIGS, fill_state_dict, make_scene_batch
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
dev = torch.device("cuda:0")
m = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
m.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(m.state_dict()).items()})
m = m.to(dev).eval()
pts, text, mask, img = make_scene_batch(cfg)
P = [torch.from_numpy(p).to(dev) for p in pts]
TD = {"text_feats": torch.from_numpy(text).to(dev), "text_token_mask": torch.from_numpy(mask).to(dev)}
IMG = torch.from_numpy(img).to(dev).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(20):
        m(P, TD, IMG)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        m(P, TD, IMG)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 200
    gs = torch.cuda.Stream(); gs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(gs):
        for _ in range(3):
            m.forward_padded(P, TD, IMG)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=gs):
            out = m.forward_padded(P, TD, IMG)
    torch.cuda.synchronize()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 200
env = {k: os.environ[k] for k in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "DEBUG_HIP_FORCE_GRAPH_QUEUES", "DEBUG_HIP_GRAPH_BATCH_SIZE") if k in os.environ}
print(f"{cfg.name}: eager {1e3 * eager:.4f} ms   graph replay {1e3 * rep:.4f} ms   {env}", flush=True)
