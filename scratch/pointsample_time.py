"""N3 (image feature -> point sampling, fusion.batch_point_sample) at the shapes of the reference's detector call
(sparse_featfusion_grounder_preshape.py:428-444): 50 views of 480 x 640 images, FPN levels of 64 / 128 / 256 / 512 channels at
strides 4 / 8 / 16 / 32, the sparse voxels of each level as points.  Per level: wall time per call (events) and the bytes that
have to move (feature maps once each way for the channels-last copy + the gathered rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from proxytransformation_amd import fusion
dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(0)
# a room of 8 x 6 x 3 m seen by V cameras on a circle looking at its centre
def cams(V):
    K = np.array([[580.0, 0, 320, 0], [0, 580.0, 240, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    P = []
    for i in range(V):
        a = 2 * np.pi * i / V
        eye = np.array([4 + 3.5 * np.cos(a), 3 + 2.5 * np.sin(a), 1.5])
        f = np.array([4.0, 3.0, 1.0]) - eye; f /= np.linalg.norm(f)
        r = np.cross(f, [0, 0, 1.0]); r /= np.linalg.norm(r)
        u = np.cross(f, r)
        R = np.stack([r, u, f])                       # world -> camera (x right, y down-ish, z forward)
        E = np.eye(4); E[:3, :3] = R; E[:3, 3] = -R @ eye
        P.append(K @ E)
    return torch.from_numpy(np.stack(P).astype(np.float32)).to(dev)
proj = cams(V)
levels = [(64, 120, 160, 100000), (128, 60, 80, 50000), (256, 30, 40, 15000), (512, 15, 20, 4000)]
tot = 0.0
for dt in (torch.float32, torch.bfloat16):
    tot = 0.0
    for C, H, W, N in levels:
        feats = torch.randn((V, C, H, W), device=dev).to(dt)
        pts = torch.from_numpy((rng.random((N, 3)) * [8, 6, 3]).astype(np.float32)).to(dev)
        for _ in range(3):
            out = fusion.batch_point_sample(None, feats, pts, proj, img_pad_shape=(480, 640), img_shape=(480, 640))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            out = fusion.batch_point_sample(None, feats, pts, proj, img_pad_shape=(480, 640), img_shape=(480, 640))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        tot += ms
        esz = feats.element_size()
        fmap = V * C * H * W
        print(f"{str(dt)[6:]:9s} level C={C:3d} {H}x{W}  N={N:6d}: {ms*1e3:7.1f} us per call; feature maps {fmap*esz/1e6:6.1f} MB in + {fmap*4/1e6:6.1f} MB channels-last copy "
              f"(= {(fmap*esz + 2*fmap*4)/1e6/ (ms*1e3) :5.2f} TB/s if that were all)")
    print(f"{str(dt)[6:]:9s} all four levels of one scene: {tot*1e3:.0f} us")
