"""r05: compute_dtype / train-eval switches of one module: step time and a gradient checksum at a fixed dropout call number."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from proxytransformation_amd import MODELS
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch
dev = torch.device("cuda:0")
cfg = PreshapeConfig("sweep", B=3, N=20000, grid_size=6, dynamic_drop_radio=0.6, L=9, V=5, seed_base=7700)
mt = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
mt.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(mt.state_dict()).items()})
mt = mt.to(dev).train()
pts, text, mask, img = make_scene_batch(cfg)
P = [torch.from_numpy(p).to(dev) for p in pts]
TD = {"text_feats": torch.from_numpy(text).to(dev).requires_grad_(True), "text_token_mask": torch.from_numpy(mask).to(dev)}
IMG = torch.from_numpy(img).to(dev).requires_grad_(True)
leaves = list(mt.parameters()) + [TD["text_feats"], IMG]
sd0 = {k: v.clone() for k, v in mt.state_dict().items()}
def step():
    for p in leaves: p.grad = None
    sum(o.sum() for o in mt(P, TD, IMG)).backward()
def probe(label):
    mt.load_state_dict(sd0)                     # running statistics back to the start
    for _ in range(20): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    mt.load_state_dict(sd0)
    mt._train_calls = 1000
    step()
    cs = float(mt.textformer[-1].mlp.fc1.weight.grad.double().abs().sum())
    print(f"{label:36s} {1e3 * dt:.3f} ms/step   checksum of fc1.weight.grad {cs:.9e}", flush=True)
probe("fp32")
probe("fp32 again")
mt.compute_dtype = "bf16"; probe("bf16 compute")
mt.compute_dtype = "fp32"; probe("fp32 after bf16")
mt.eval()
with torch.no_grad():
    for _ in range(50): mt(P, TD, IMG)
mt.train(); probe("fp32 after an eval phase")
