"""Which products of a training step go through the generic strided kernel (ptx_op_gemm = k_bgemm): shapes, strides, slices,
and the time of each shape alone (events around 20 repeats): python scratch/train_gemm_shapes.py"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxytransformation_amd import MODELS, train
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch
cfg = PreshapeConfig("cfg4train", B=6, N=100000, grid_size=12, dynamic_drop_radio=0.6, L=20, V=20, text_blocks=3, img_blocks=3, seed_base=4500)
m = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
m.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(m.state_dict()).items()})
m = m.cuda().train()
pts, text, mask, img = make_scene_batch(cfg)
dev = torch.device("cuda:0")
args = ([torch.from_numpy(p).to(dev) for p in pts], {"text_feats": torch.from_numpy(text).to(dev).requires_grad_(True),
        "text_token_mask": torch.from_numpy(mask).to(dev)}, torch.from_numpy(img).to(dev).requires_grad_(True))
leaves = list(m.parameters()) + [args[1]["text_feats"], args[2]]
def step():
    for t in leaves: t.grad = None
    outs = m(*args)
    sum(o.sum() for o in outs).backward()
for _ in range(2): step()
torch.cuda.synchronize()
orig = train.gemm
calls = collections.OrderedDict()
def logged(A, B, C, M, N, K, **kw):
    key = (M, N, K, kw.get("a"), kw.get("b"), kw.get("batch", 1), kw.get("inner", 1), kw.get("ksplit", 1), kw.get("a_dtype", 0), kw.get("b_dtype", 0), bool(kw.get("accumulate", False)))
    if key not in calls:
        calls[key] = [0, (A, B, C, M, N, K, dict(kw))]
    calls[key][0] += 1
    return orig(A, B, C, M, N, K, **kw)
train.gemm = logged
step()
torch.cuda.synchronize()
train.gemm = orig
tot = 0.0
print(f"{'M':>6} {'N':>6} {'K':>7} a_strides  b_strides  batch inner ksplit adt bdt acc | calls   us/call   us/step")
for key, (n, (A, B, C, M, N, K, kw)) in calls.items():
    for _ in range(3): orig(A, B, C, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): orig(A, B, C, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    tot += us * n
    print(f"{M:6d} {N:6d} {K:7d} {str(key[3]):>10} {str(key[4]):>10} {key[5]:5d} {key[6]:5d} {key[7]:6d} {key[8]:3d} {key[9]:3d} {int(key[10]):3d} | {n:5d} {us:9.1f} {us * n:9.1f}")
print(f"sum over the step: {tot:.0f} us")
