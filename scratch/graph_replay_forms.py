"""r05: HIP-graph replay of forward_padded -- one graph replayed, three graphs (one per input set) alternating, one graph holding three
forwards.  Per-forward time; eager beside it (same rotation of input sets)."""
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
sets = []
for j in range(3):
    pts, text, mask, img = make_scene_batch(cfg, scene_ids=range(j * cfg.B, (j + 1) * cfg.B))
    sets.append(([torch.from_numpy(p).to(dev) for p in pts],
                 {"text_feats": torch.from_numpy(text).to(dev), "text_token_mask": torch.from_numpy(mask).to(dev)},
                 torch.from_numpy(img).to(dev).to(torch.bfloat16)))
N = 240


def timed(fn, per):
    for _ in range(12):
        fn(_)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N // per):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / (N // per * per)


with torch.no_grad():
    print(f"eager, three input sets in rotation      {timed(lambda i: m(*sets[i % 3]), 1):.4f} ms per forward")
    gs = torch.cuda.Stream(); gs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(gs):
        for j in range(3):
            m.forward_padded(*sets[j])
    torch.cuda.synchronize()
    graphs = []
    for j in range(3):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=gs):
            r = m.forward_padded(*sets[j])
        graphs.append((g, r))
    print(f"one graph replayed                        {timed(lambda i: graphs[0][0].replay(), 1):.4f} ms per forward")
    print(f"three graphs alternating                  {timed(lambda i: graphs[i % 3][0].replay(), 1):.4f} ms per forward")
    with torch.cuda.stream(gs):
        print(f"three graphs alternating, on the capture stream {timed(lambda i: graphs[i % 3][0].replay(), 1):.4f} ms per forward")
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3, stream=gs):
        r3 = [m.forward_padded(*sets[j]) for j in range(3)]
    print(f"one graph of three forwards               {timed(lambda i: g3.replay(), 3):.4f} ms per forward")
    for reps in (2, 4):
        gk = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gk, stream=gs):
            rk = [m.forward_padded(*sets[j % 3]) for j in range(3 * reps)]
        print(f"one graph of {3 * reps:2d} forwards                  {timed(lambda i: gk.replay(), 3 * reps):.4f} ms per forward")
        del gk, rk
    # the result of the last form equals the eager one
    g3.replay(); torch.cuda.synchronize()
    for j in range(3):
        outs = m(*sets[j])
        out, counts = r3[j]
        for b in range(cfg.B):
            assert torch.equal(out[b, : int(counts[b])], outs[b])
    print("graph of three forwards == eager: ok")
