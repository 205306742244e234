"""r05: does the training step stay at its speed over a long run?  N steps at the training shape, a line per 100 steps:
ms/step, allocator state, python object count.  argv: steps [side 0/1] [apart 0/1] [one_node 0/1]"""
import gc, sys, time
import os; sys.path.insert(0, os.environ.get("PTX_PKG_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from proxytransformation_amd import MODELS, train
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
train._SIDE_STREAM = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
train._BLOCKS_APART = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
train._ONE_NODE = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
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


print(f"side={train._SIDE_STREAM} apart={train._BLOCKS_APART} one_node={train._ONE_NODE}")
for _ in range(5):
    step()
torch.cuda.synchronize()


class TimedEvent(torch.cuda.Event):
    waited = 0.0

    def synchronize(self):
        t = time.perf_counter()
        super().synchronize()
        TimedEvent.waited += time.perf_counter() - t


for k, (buf, ev) in list((mod._train_pin or {}).items()):      # (the Python-bodied node's pinned words; the C step keeps its own)
    mod._train_pin[k] = (buf, TimedEvent())
GC = {"t": 0.0, "n": [0, 0, 0], "t0": 0.0}


def _gc_cb(phase, info):
    if phase == "start":
        GC["t0"] = time.perf_counter()
    else:
        GC["t"] += time.perf_counter() - GC["t0"]
        GC["n"][info["generation"]] += 1


gc.callbacks.append(_gc_cb)
if os.environ.get("GC_FREEZE"):
    gc.collect(); gc.freeze()
for blk in range(steps // 100):
    t0 = time.perf_counter()
    TimedEvent.waited = 0.0
    GC["t"], GC["n"] = 0.0, [0, 0, 0]
    for _ in range(100):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms = torch.cuda.memory_stats()
    print(f"steps {blk * 100:5d}+: {10 * (t2 - t0):7.3f} ms/step (host {10 * (t1 - t0):7.3f}, of which waiting for the counts {10 * TimedEvent.waited:6.3f}, in the collector {10 * GC['t']:6.3f}: {GC['n']} runs)  allocated {ms['allocated_bytes.all.current'] >> 20} MiB "
          f"reserved {ms['reserved_bytes.all.current'] >> 20} MiB  segments {ms['segment.all.current']}  live blocks {ms['allocation.all.current']} "
          f"inactive-split {ms['inactive_split.all.current']}  gos {len(gos)}  py objects {len(gc.get_objects())}", flush=True)
