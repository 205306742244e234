"""r05: every public entry point called a few hundred times -- device memory (allocator's live bytes / blocks), reserved memory and
the process's resident set must stop moving after the warm-up calls.  Prints one line per entry point."""
import gc, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import psutil
import torch
from proxytransformation_amd import MODELS
from proxytransformation_amd.ingest import MultiViewIngest
from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch

dev = torch.device("cuda:0")
proc = psutil.Process()
cfg = PreshapeConfig("sweep", B=3, N=20000, grid_size=6, dynamic_drop_radio=0.6, L=9, V=5, seed_base=7700)


def build(train=False):
    m = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(m.state_dict()).items()})
    return m.to(dev).train(train)


pts, text, mask, img = make_scene_batch(cfg)
P = [torch.from_numpy(p).to(dev) for p in pts]
TD = {"text_feats": torch.from_numpy(text).to(dev), "text_token_mask": torch.from_numpy(mask).to(dev)}
IMG = torch.from_numpy(img).to(dev)
IMG16 = IMG.to(torch.bfloat16)


def state():
    torch.cuda.synchronize()
    s = torch.cuda.memory_stats()
    return s["allocated_bytes.all.current"], s["allocation.all.current"], s["reserved_bytes.all.current"], proc.memory_info().rss


def sweep(name, fn, warm=40, calls=400):
    for _ in range(warm):
        fn()
    gc.collect()
    a = state()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    dt = (time.perf_counter() - t0) / calls
    gc.collect()
    b = state()
    verdict = "ok" if (a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and b[3] - a[3] < (8 << 20)) else "MOVED"
    print(f"{name:44s} {1e3 * dt:7.3f} ms/call  device bytes {b[0] - a[0]:+d}  blocks {b[1] - a[1]:+d}  reserved {b[2] - a[2]:+d}  "
          f"host rss {(b[3] - a[3]) / 1048576:+.1f} MiB  {verdict}", flush=True)


with torch.no_grad():
    m = build()
    sweep("eval forward, fp32 features", lambda: m(P, TD, IMG))
    sweep("eval forward, bf16 features", lambda: m(P, TD, IMG16))
    sweep("eval forward, return_transforms", lambda: m(P, TD, IMG16, return_transforms=True))
    sweep("forward_debug", lambda: m.forward_debug(P, TD, IMG16), calls=200)
    sweep("forward_padded", lambda: m.forward_padded(P, TD, IMG16))
    sweep("forward + quantize", lambda: m.quantize(m(P, TD, IMG16), 0.02))
    sweep("forward + quantize(return_inverse)", lambda: m.quantize(m(P, TD, IMG16), 0.02, return_inverse=True))
    s2 = torch.cuda.Stream()

    def two_streams():
        m(P, TD, IMG16)
        with torch.cuda.stream(s2):
            m(P, TD, IMG16)
    s2.wait_stream(torch.cuda.current_stream())
    sweep("eval forward on two streams", two_streams)
    # graph replay
    gs = torch.cuda.Stream()
    gs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(gs):
        for _ in range(3):
            m.forward_padded(P, TD, IMG16)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=gs):
            out = m.forward_padded(P, TD, IMG16)
    torch.cuda.synchronize()
    sweep("HIP graph replay of forward_padded", lambda: g.replay())
    # a module built, used and dropped again and again (contexts, workspaces, pinned words go with it)
    def build_use_drop():
        mm = build()
        mm(P, TD, IMG16)
        mm.close()
    sweep("build + forward + close", build_use_drop, warm=5, calls=30)
    # ingest
    rng = np.random.default_rng(7)
    scenes = []
    for b in range(2):
        V, H, W = 3, 96, 128
        depth = (2.0 + 6.0 * rng.random((V, H, W))).astype(np.float32)
        depth[rng.random((V, H, W)) < 0.2] = 0.0
        K = np.array([[60.0, 0, 64], [0, 60.0, 48], [0, 0, 1]])
        ext = np.stack([np.eye(4, dtype=np.float32) for _ in range(V)])
        scenes.append(dict(depth_img=torch.from_numpy(depth).to(dev), depth_cam2img=K, extrinsic=ext))
    ing = MultiViewIngest(6000)
    r = np.random.RandomState(3)
    sweep("MultiViewIngest", lambda: ing(scenes, rng=r))

mt = build(train=True)
TDg = {"text_feats": TD["text_feats"].clone().requires_grad_(True), "text_token_mask": TD["text_token_mask"]}
IMGg = IMG.clone().requires_grad_(True)
leaves = list(mt.parameters()) + [TDg["text_feats"], IMGg]


def train_step(transforms=False):
    for p in leaves:
        p.grad = None
    if transforms:
        outs, tr = mt(P, TDg, IMGg, return_transforms=True)
        (sum(o.sum() for o in outs) + sum(v.square().mean() for v in tr.values())).backward()
    else:
        sum(o.sum() for o in mt(P, TDg, IMGg)).backward()


sweep("train step", train_step, calls=200)
sweep("train step, return_transforms", lambda: train_step(True), calls=200)
mt.compute_dtype = "bf16"
sweep("train step, bf16 compute", train_step, calls=200)
mt.compute_dtype = "fp32"
mt.eval()
with torch.no_grad():
    sweep("eval forward of a module that trained", lambda: mt(P, TD, IMG16))
mt.train()
sweep("train step after an eval phase", train_step, calls=100)
m.check(); mt.check()
