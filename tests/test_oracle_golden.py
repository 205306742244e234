"""Pin the CPU oracle (oracle/) against the golden vectors captured from the reference file
itself (tests/golden/gen_golden.py).  Integer results must be bit-identical; float results
are compared at tolerances far below the 1e-4 coordinate tolerance of the north star."""
import numpy as np
import pytest

from oracle import oracle
from tests.util import GOLDEN_CASES, assert_close, build_module, golden_cfg, load_golden, oracle_kwargs


def _run(name, mode, inject_centers=True):
    g = load_golden(name)
    cfg = golden_cfg(g)
    _, sd = build_module(cfg)
    out = oracle.forward(
        sd, **oracle_kwargs(cfg), points=g["points"], text_feats=g["text_feats"],
        text_mask=g["text_mask"], img_feat=g["img_feat"],
        order_override=g["order_shipped"] if mode == "shipped" else None,
        centers_override=g["centers"] if inject_centers else None, num_threads=1)
    return g, cfg, out


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_clustering_matches_reference(name):
    g, cfg, out = _run(name, "stable", inject_centers=False)
    # grid centres are bit-exact, gathered clusters are copies
    assert np.array_equal(out["centers0"], g["centers0"])
    assert np.array_equal(out["cluster1"], g["cluster1"])
    # offset net: same math, different summation order
    assert_close(out["centers"], g["centers"], atol=2e-5, what="clamped centres")
    # golden scenes are boundary-safe (margin recorded by the generator), so membership is identical
    assert g["bq2_boundary_margin"].min() > 1e-4
    assert np.array_equal(out["idx2"], g["idx2"])
    assert np.array_equal(out["cluster2"], g["cluster2"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("mode", ["stable", "shipped"])
def test_selection_and_outputs_match_reference(name, mode):
    g, cfg, out = _run(name, mode)
    assert np.array_equal(out["idx2"], g["idx2"])
    assert np.array_equal(out["pad_counts"], g["pad_counts"])
    assert np.array_equal(out["order"], g[f"order_{mode}"])          # stable counting sort == torch stable argsort
    assert np.array_equal(out["picks"], g[f"fps_{mode}"])            # FPS == PRE's sample_farthest_points_naive
    assert np.array_equal(out["kidx"], g[f"kidx_{mode}"])
    assert np.array_equal(out["drop_idx"], g[f"drop_idx_{mode}"])
    assert_close(out["translate"], g[f"translate_{mode}"], atol=2e-5, rtol=1e-5, what="translate")
    assert_close(out["transform"], g[f"transform_{mode}"], atol=2e-5, rtol=1e-5, what="transform")
    assert_close(out["new_points"], g[f"new_points_{mode}"], atol=5e-5, what="new_points")
    for b in range(cfg.B):
        ref = g[f"out_{mode}_{b}"]
        assert out["outputs"][b].shape == ref.shape                  # same points dropped
        assert_close(out["outputs"][b], ref, atol=5e-5, what=f"output {b}")


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_float_stages_match_reference(name):
    g, cfg, out = _run(name, "stable")
    assert np.array_equal(out["kcluster"], g["kcluster_stable"])
    assert np.array_equal(out["kcenter"], g["kcenter_stable"])
    assert_close(out["point_proxy"], g["point_proxy_stable"], atol=1e-5, rtol=1e-5, what="point_proxy")
    assert_close(out["img_proxy"].reshape(g["img_proxy"].shape), g["img_proxy"], atol=2e-5, rtol=1e-5,
                 what="img_proxy")
    assert_close(out["text_block"]["out"].numpy(), g["text_block_stable"], atol=5e-5, rtol=1e-5, what="text block")
    assert_close(out["img_block"]["out"].numpy(), g["img_block_stable"], atol=5e-5, rtol=1e-5, what="img block")
    assert_close(out["text_guide"], g["text_guide_stable"], atol=2e-5, rtol=1e-5, what="text_guide")
    assert_close(out["img_guide"], g["img_guide_stable"], atol=2e-5, rtol=1e-5, what="img_guide")
    assert_close(out["new_clusters"], g["new_clusters_stable"], atol=5e-5, what="new_clusters")


def test_unstable_argsort_really_differs():
    """the as-shipped permutation is not the stable one, so both modes are exercised."""
    g = load_golden("g1_cfg1")
    assert not np.array_equal(g["order_stable"], g["order_shipped"])


def test_voxel_restatement_keeps_the_first_point_of_every_voxel():
    """oracle.voxelize (parity unpinned against MinkowskiEngine): floor(p / size) in fp32, scene index in
    front, first point of a voxel in (scene, point) order survives, inverse maps consistent."""
    a = np.array([[0.004, 0.0, 0.0], [0.013, 0.0, 0.0], [0.009, 0.0, 0.0], [-0.001, 0.5, 0.0]], np.float32)
    b = np.array([[0.004, 0.0, 0.0], [0.004, 0.0, 0.0]], np.float32)
    c, f, inv = oracle.voxelize([a, b], 0.01)
    assert c.tolist() == [[0, 0, 0, 0], [0, 1, 0, 0], [0, -1, 50, 0], [1, 0, 0, 0]]
    assert np.array_equal(f, np.stack([a[0], a[1], a[3], b[0]]))
    assert inv[0].tolist() == [0, 1, 0, 2] and inv[1].tolist() == [3, 3]


def _meta3d(g):
    return dict(transformation_3d_flow=["HF", "R", "S", "T"], pcd_horizontal_flip=True, pcd_vertical_flip=False,
                pcd_rotation=g["flow3d_rot_T"], pcd_scale_factor=float(g["flow3d_scale"]), pcd_trans=g["flow3d_trans"])


@pytest.mark.parametrize("case", ["plain", "aug", "flow3d", "bilinear"])
def test_point_sample_restatement_matches_the_reference(case):
    """oracle.point_sample against batch_point_sample run from the reference file (tests/golden/g5_point_sample.npz;
    SURVEY 8f N3): same pixels, same valid-view counts -> identical up to the fp32 order of the view sum."""
    g = load_golden("g5_point_sample")
    sx, sy, cw, ch, flip, ori_w = [float(x) for x in g[f"{case}_cfg"]]
    pts = g[f"{case}_points"]
    if case == "flow3d":            # the training pipeline's 3D augmentation, undone as apply_3d_transformation(reverse=True) does
        pts = oracle.reverse_3d_points(pts, _meta3d(g))
    out, nvalid = oracle.point_sample(pts, g["feats"], g["proj"], scale=(sx, sy), crop=(cw, ch),
                                      flip=bool(flip), ori_w=ori_w, pad_hw=tuple(float(x) for x in g["pad"]),
                                      bilinear=case == "bilinear")
    assert_close(out, g[f"{case}_out"], atol=1e-5, rtol=1e-5, what="sampled features")
    assert (nvalid > 0).sum() == (np.abs(g[f"{case}_out"]).sum(1) > 0).sum()


@pytest.mark.parametrize("ct", ["DEPTH", "LIDAR", "CAMERA"])
@pytest.mark.parametrize("fl", ["hf", "vf"])
def test_reverse_3d_flow_per_coordinate_type(ct, fl):
    """The reverse 3D augmentation flow per coordinate type and flip against the reference's own
    apply_3d_transformation(reverse=True) (g5_point_sample: rev3d_*): LiDARPoints.flip negates y for 'horizontal' and x for
    'vertical', the opposite of DepthPoints, CameraPoints x / z -- the oracle's step-by-step form and the product's
    composed (3,4) affine (proxytransformation_amd.fusion.reverse_3d_flow, host arithmetic) must both follow."""
    from proxytransformation_amd.fusion import reverse_3d_flow
    g = load_golden("g5_point_sample")
    meta = dict(_meta3d(g), transformation_3d_flow=["VF", "HF", "R", "S", "T"], pcd_horizontal_flip=fl == "hf",
                pcd_vertical_flip=fl == "vf")
    pts, want = g["rev3d_points"], g[f"rev3d_{ct}_{fl}"]
    assert_close(oracle.reverse_3d_points(pts, meta, ct), want, atol=2e-6, what=f"oracle reverse flow ({ct}, {fl})")
    A = reverse_3d_flow(meta, ct).numpy().astype(np.float64)
    assert_close(pts.astype(np.float64) @ A[:, :3].T + A[:, 3], want, atol=5e-6, what=f"composed reverse flow ({ct}, {fl})")
    if ct == "LIDAR":
        assert np.abs(want - g[f"rev3d_DEPTH_{fl}"]).max() > 0.1    # the types really differ


# ------------------------------------------------------------------ multi-view depth ingest
@pytest.mark.parametrize("case", ["plain", "aug"])
def test_ingest_restatement_matches_the_reference_transforms(case):
    """oracle.ingest against tests/golden/g6_ingest.npz, which gen_golden.py captured by running the reference's own
    MultiViewPipeline(ConvertRGBDToPoints, PointSample) -> AggregateMultiViewPoints -> PointSample (-> GlobalRotScaleTrans)
    with np.random.seed fixed: the same seed must pick the same pixels (the restatement consumes the RNG in the
    reference's order) and give the same coordinates."""
    g = load_golden("g6_ingest")
    depth = g["depth_u16"].astype(np.float32) / float(g["depth_shift"])
    aug = None
    np.random.seed(int(g[f"{case}_seed"]))
    if case == "aug":
        aug = dict(rot_mat_T=g["aug_rot_mat_T"], scale=g["aug_scale"], trans=g["aug_trans"])
    out = oracle.ingest(depth, g["depth_cam2img"], g["extrinsic"], int(g["n_points"]), rng=np.random, aug=aug)
    assert out["view_counts"][3] < int(g["n_points"]) // 10          # one view is sampled with replacement
    assert_close(out["points"], g[f"{case}_points"], atol=2e-6, what=f"g6_ingest/{case}")
