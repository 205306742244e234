"""Image feature -> point sampling on the GPU against the capture from the reference's own
batch_point_sample (tests/golden/g5_point_sample.npz) and against the CPU restatement on fresh data."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, load_golden

pytestmark = pytest.mark.gpu


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("case", ["plain", "aug", "flow3d", "bilinear"])
def test_point_sample_matches_the_reference_capture(case):
    """plain / aug: image-side augmentation only; flow3d: img_meta carries the training pipeline's 3D augmentation
    (RandomFlip3D + GlobalRotScaleTrans: 'HF', 'R', 'S', 'T'), undone inside the call like apply_3d_transformation(reverse=True)
    does; bilinear: aligned=True."""
    from proxytransformation_amd.fusion import batch_point_sample
    from tests.test_oracle_golden import _meta3d
    g = load_golden("g5_point_sample")
    sx, sy, cw, ch, flip, ori_w = [float(x) for x in g[f"{case}_cfg"]]
    meta = _meta3d(g) if case == "flow3d" else {}
    if meta:
        meta["pcd_rotation"] = torch.from_numpy(meta["pcd_rotation"])          # the pipeline stores a tensor
    out = batch_point_sample(meta, _t(g["feats"]), _t(g[f"{case}_points"]), _t(g["proj"]), "DEPTH",
                             img_scale_factor=torch.tensor([sx, sy]), img_crop_offset=torch.tensor([cw, ch]),
                             img_flip=bool(flip), img_pad_shape=(int(g["pad"][0]), int(g["pad"][1])),
                             img_shape=(600, int(ori_w)), aligned=case == "bilinear")
    assert_close(out.cpu().numpy(), g[f"{case}_out"], atol=1e-5, rtol=1e-5, what="sampled features")


def test_reverse_flow_composition_on_the_host():
    """reverse_3d_flow == the reference's step-by-step reverse flow (oracle.reverse_3d_points), for every op and order."""
    from oracle import oracle
    from proxytransformation_amd.fusion import reverse_3d_flow
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((200, 3)).astype(np.float32) * 4
    a = 0.3
    meta = dict(transformation_3d_flow=["T", "VF", "S", "R", "HF", "T"], pcd_horizontal_flip=True, pcd_vertical_flip=True,
                pcd_rotation=np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32),
                pcd_scale_factor=0.93, pcd_trans=np.array([0.5, -0.25, 0.125], np.float32))
    A = reverse_3d_flow(meta).numpy().astype(np.float64)
    got = pts.astype(np.float64) @ A[:, :3].T + A[:, 3]
    assert_close(got, oracle.reverse_3d_points(pts, meta), atol=2e-6, what="reverse flow")
    assert reverse_3d_flow({}) is None and reverse_3d_flow(None) is None


@pytest.mark.parametrize("dtype,V,C", [(torch.float32, 50, 256), (torch.bfloat16, 70, 96), (torch.float16, 3, 512)])
def test_point_sample_matches_the_restatement(dtype, V, C):
    """More views than lanes (70), channel counts that are no multiple of 64, 16-bit feature maps, points behind the
    cameras and outside every image; bit-identical to the restatement (same fp32 operation order)."""
    from oracle import oracle
    from proxytransformation_amd.fusion import batch_point_sample
    rng = np.random.default_rng(V)
    H, W, N = 17, 23, 5000
    feats = torch.from_numpy(rng.standard_normal((V, C, H, W), dtype=np.float32)).to(dtype)
    proj = np.zeros((V, 4, 4), np.float32)
    for v in range(V):
        ang = 2 * np.pi * v / V
        ext = np.eye(4)
        ext[:3, :3] = [[np.cos(ang), 0, -np.sin(ang)], [0, 1, 0], [np.sin(ang), 0, np.cos(ang)]]
        ext[:3, 3] = [0.1 * v - 1.0, 0.2, 3.0]
        K = np.eye(4); K[0, 0] = K[1, 1] = 300.0; K[0, 2] = 320.0; K[1, 2] = 240.0
        proj[v] = (K @ ext).astype(np.float32)
    pts = ((rng.random((N, 3)) - 0.5) * 14).astype(np.float32)
    kw = dict(scale=(0.95, 1.05), crop=(3.0, 5.0), flip=True, ori_w=640.0, pad_hw=(480.0, 640.0))
    ref, nvalid = oracle.point_sample(pts, feats.float().numpy(), proj, **kw)
    out = batch_point_sample(None, feats.cuda(), _t(pts), _t(proj), "DEPTH", img_scale_factor=(0.95, 1.05),
                             img_crop_offset=(3.0, 5.0), img_flip=True, img_pad_shape=(480, 640), img_shape=(480, 640))
    assert (nvalid == 0).any() and (nvalid >= min(3, V)).any()
    assert np.array_equal(out.cpu().numpy(), ref)
    # bilinear (aligned=True): neighbours partly outside the map, points outside every image
    ref2, _ = oracle.point_sample(pts, feats.float().numpy(), proj, bilinear=True, **kw)
    out2 = batch_point_sample(None, feats.cuda(), _t(pts), _t(proj), "DEPTH", img_scale_factor=(0.95, 1.05),
                              img_crop_offset=(3.0, 5.0), img_flip=True, img_pad_shape=(480, 640), img_shape=(480, 640),
                              aligned=True)
    assert_close(out2.cpu().numpy(), ref2, atol=2e-6, rtol=1e-6, what="bilinear samples")
