"""Throughput of the N4 ingest at the reference's shapes (50 views of 480 x 640 uint16 depth -> 100k points per scene):
python scratch/ingest_time.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
from proxytransformation_amd import _abi
from proxytransformation_amd.ingest import MultiViewIngest, compose_choices, lu_factor_4x4
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
V, H, W, N = 50, 480, 640, 100000
rng = np.random.default_rng(0)
dev = torch.device("cuda:0")
scenes = []
for b in range(B):
    d = (500 + 4000 * rng.random((V, H, W))).astype(np.uint16)
    d[rng.random((V, H, W)) < 0.15] = 0
    K = np.array([[577.0, 0, 319.5], [0, 577.0, 239.5], [0, 0, 1]])
    ext = np.stack([np.eye(4, dtype=np.float32) for _ in range(V)])
    for v in range(V):
        a = 0.13 * v
        ext[v, :3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
        ext[v, :3, 3] = [0.05 * v, -0.03 * v, 1.2]
    scenes.append(dict(depth_img=torch.from_numpy(d.view(np.int16)).to(dev).view(torch.uint16), depth_shift=1000.0, depth_cam2img=K, extrinsic=ext))
ing = MultiViewIngest(N)
r = np.random.RandomState(1)
for _ in range(2): batch = ing(scenes, rng=r)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): batch = ing(scenes, rng=r)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"MultiViewIngest B={B}: {1e3*dt:.2f} ms per batch ({1e3*dt/B:.2f} ms per scene) incl. host RNG / composition / uploads")
# kernels alone (events): index pass and gather of one scene
lib = _abi.lib(); st = torch.cuda.current_stream().cuda_stream
depth = scenes[0]["depth_img"]
nbytes = lib.ptx_ingest_workspace_bytes(V, H, W)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev); counts = torch.empty(V, dtype=torch.int32).pin_memory()
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
reps = 20
lib.ptx_ingest_index(depth.data_ptr(), 1, V, H, W, ws.data_ptr(), nbytes, counts.data_ptr(), st); torch.cuda.synchronize()
e[0].record()
for _ in range(reps): lib.ptx_ingest_index(depth.data_ptr(), 1, V, H, W, ws.data_ptr(), nbytes, counts.data_ptr(), st)
e[1].record(); torch.cuda.synchronize()
t_idx = e[0].elapsed_time(e[1]) / reps
sel = torch.from_numpy(compose_choices(counts.numpy(), N // 10, N, np.random.RandomState(2))).to(dev)
inv_k = torch.from_numpy(ing._intrinsics(scenes[0]["depth_cam2img"], V)).to(dev)
lus, pivs = zip(*(lu_factor_4x4(scenes[0]["extrinsic"][v]) for v in range(V)))
lu = torch.from_numpy(np.stack(lus)).to(dev); piv = torch.from_numpy(np.stack(pivs)).to(dev)
out = torch.empty((N, 3), device=dev); bbox = torch.empty(6, dtype=torch.int32, device=dev); status = torch.zeros(1, dtype=torch.int32, device=dev)
def gather():
    lib.ptx_ingest_gather(depth.data_ptr(), 1, 1000.0, V, H, W, inv_k.data_ptr(), lu.data_ptr(), piv.data_ptr(), sel.data_ptr(), N, None,
                          out.data_ptr(), bbox.data_ptr(), status.data_ptr(), ws.data_ptr(), nbytes, st)
gather(); torch.cuda.synchronize()
e[2].record()
for _ in range(reps): gather()
e[3].record(); torch.cuda.synchronize()
t_g = e[2].elapsed_time(e[3]) / reps
mb = V * H * W * 2 / 1e6
print(f"k_ingest_index + scan: {1e3*t_idx:.1f} us for {mb:.1f} MB of uint16 depth = {mb/1e3/t_idx*1e3/1e3:.2f} TB/s; k_ingest_gather (N={N}): {1e3*t_g:.1f} us")
