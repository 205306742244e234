"""r05 A/B: the one-node training step with the image block on the side stream beside the text block (_BLOCKS_APART) vs both
blocks on the caller's stream.  Interleaved rounds of 30 steps each at the reference's training shape (bench.py --train)."""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from proxytransformation_amd import MODELS, train
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch

dev = torch.device("cuda:0")
cfg = PreshapeConfig("cfg4train", B=6, N=100000, grid_size=12, dynamic_drop_radio=0.6, L=20, V=20, text_blocks=3, img_blocks=3,
                     seed_base=4500)
mod = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
mod.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(mod.state_dict()).items()})
mod = mod.to(dev).train()
pts, text, mask, img = make_scene_batch(cfg)
args = ([torch.from_numpy(p).to(dev) for p in pts],
        {"text_feats": torch.from_numpy(text).to(dev).requires_grad_(True), "text_token_mask": torch.from_numpy(mask).to(dev)},
        torch.from_numpy(img).to(dev).requires_grad_(True))
leaves = list(mod.parameters()) + [args[1]["text_feats"], args[2]]
gos = {}


def step():
    for t in leaves:
        t.grad = None
    outs = mod(*args)
    key = tuple(o.shape[0] for o in outs)
    if key not in gos:
        gos[key] = [torch.ones_like(o) for o in outs]
    torch.autograd.backward(outs, gos[key])


def timed(n=30):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


res = {True: [], False: []}
for r in range(16):
    for flag in (True, False) if r % 2 == 0 else (False, True):
        train._BLOCKS_APART = flag
        res[flag].append(timed())
for flag in (True, False):
    v = sorted(res[flag])
    print(f"_BLOCKS_APART={flag}: median {v[len(v) // 2]:.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}  ({' '.join(f'{x:.3f}' for x in res[flag])})")
# gradients agree between the two forms (same seeds => same dropout masks: fix the call counter)
def grads(flag):
    train._BLOCKS_APART = flag
    mod._train_calls = 7
    step()
    torch.cuda.synchronize()
    return [None if t.grad is None else t.grad.clone() for t in leaves]
ga, gb = grads(True), grads(False)
worst = 0.0
for a, b in zip(ga, gb):
    if a is None or b is None:
        assert a is None and b is None
        continue
    worst = max(worst, float((a - b).abs().max() / (b.abs().max() + 1e-30)))
print(f"largest relative gradient difference between the two forms: {worst:.3e}")
