"""BASELINE configs[3] as a PIPELINE (VERDICT r05 "next" #1): synthetic multi-view depth maps -> MultiViewIngest (N4) -> the neck
with the shipped configuration's weights and the ingested bounding boxes -> module.quantize (N2, 1 cm voxels) -> the coordinates of
MinkResNet's four output levels -> fusion.batch_point_sample on the four 2D-backbone levels (N3), chained by
``pipeline.GroundingFeaturePrefix`` on the caller's stream -- the channels-last copies of the feature maps run on a side stream beside
the ingest and are joined by an event -- (detectors/sparse_featfusion_grounder_preshape.py:385-448,
configs/grounding/proxy-tiblock33-gs12-wbias-ddr0.6-clip.py:105-142).

Held against the oracle chain STAGE BY STAGE -- every oracle stage consumes what the GPU's previous stage produced, so that a float
difference of 1e-5 m in one stage cannot flip an integer decision of the next and every index tensor / voxel row / sampled feature
can be required bit-identical -- at a reduced size (V = 8 views, N = 20 000) for every scene and at the shipped size (50 views of
480 x 640, N = 100 000, gs = 12, six scenes) for two scenes, plus size-independent properties for all six; the chained call may not
synchronise the device anywhere (the Python-level synchronise entry points are counted)."""
import contextlib

import numpy as np
import pytest
import torch

from proxytransformation_amd.pipeline import MINK_RESNET_STRIDES, GroundingFeaturePrefix, projection_matrices
from proxytransformation_amd.synth import CONFIGS, FPN_LEVELS, PreshapeConfig, make_depth_scene
from tests.util import assert_close, build_module, oracle_kwargs

pytestmark = pytest.mark.gpu

INT_KEYS = ("idx2", "order", "picks", "keep", "kidx", "drop_idx")


def _dev():
    return torch.device("cuda:0")


def _inputs(cfg, n_scenes, V, seed0, as_u16=True):
    """Scenes (numpy + device twins), text proxies and the four feature levels (B,V,C_l,s_l,s_l) generated ON the device."""
    dev = _dev()
    scenes_np = [make_depth_scene(seed0 + b, V=V, as_u16=as_u16) for b in range(n_scenes)]
    scenes = []
    for sc in scenes_np:
        d = sc["depth_img"]
        dt = torch.from_numpy(d.view(np.int16)).to(dev).view(torch.uint16) if d.dtype == np.uint16 else torch.from_numpy(d).to(dev)
        scenes.append(dict(sc, depth_img=dt))
    g = torch.Generator(device=dev)
    g.manual_seed(seed0)
    feats = [torch.randn((n_scenes, V, c, s, s), generator=g, device=dev, dtype=torch.float32) for c, s in FPN_LEVELS]
    text = torch.randn((n_scenes, cfg.L, cfg.embed_dim), generator=g, device=dev)
    mask = torch.ones((n_scenes, cfg.L), dtype=torch.bool, device=dev)
    if n_scenes > 1:
        mask[1, cfg.L - cfg.L // 3:] = False
    return scenes_np, scenes, {"text_feats": text, "text_token_mask": mask}, feats


@contextlib.contextmanager
def _count_synchronises():
    """Every Python-level way to drain the device / a stream / an event, counted."""
    calls = []
    saved = (torch.cuda.synchronize, torch.cuda.Stream.synchronize, torch.cuda.Event.synchronize)

    def wrap(name, fn):
        def inner(*a, **k):
            calls.append(name)
            return fn(*a, **k)
        return inner
    torch.cuda.synchronize = wrap("torch.cuda.synchronize", saved[0])
    torch.cuda.Stream.synchronize = wrap("Stream.synchronize", saved[1])
    torch.cuda.Event.synchronize = wrap("Event.synchronize", saved[2])
    try:
        yield calls
    finally:
        torch.cuda.synchronize, torch.cuda.Stream.synchronize, torch.cuda.Event.synchronize = saved


def _run_chain(cfg, n_scenes, V, seed0, as_u16=True):
    m, sd = build_module(cfg)
    m = m.cuda()
    scenes_np, scenes, text_dict, feats = _inputs(cfg, n_scenes, V, seed0, as_u16)
    pipe = GroundingFeaturePrefix(m, n_points=cfg.N)
    pipe(scenes, text_dict, feats, rng=np.random.RandomState(seed0))                 # warm-up: lanes, workspaces, parameter tables
    torch.cuda.synchronize()
    with _count_synchronises() as calls:
        res = pipe(scenes, text_dict, feats, rng=np.random.RandomState(seed0))
    assert calls == [], f"the chained call synchronised: {calls}"
    torch.cuda.synchronize()
    m.check()
    return m, sd, pipe, scenes_np, scenes, text_dict, feats, res


def _check_stagewise(cfg, m, sd, scenes_np, text_dict, feats, res, seed0, which):
    """Oracle stage k on the GPU's stage k-1, for the scenes ``which``."""
    from oracle import oracle
    rs = np.random.RandomState(seed0)
    refs = {}
    for b, sc in enumerate(scenes_np):                                    # the host RNG stream runs over ALL scenes in order
        if b > max(which):
            break
        d = sc["depth_img"]
        depth = d.astype(np.float32) / np.float32(sc["depth_shift"]) if d.dtype == np.uint16 else d
        refs[b] = oracle.ingest(depth, sc["depth_cam2img"], sc["extrinsic"], cfg.N, rng=rs)
    text = text_dict["text_feats"].cpu().numpy()
    mask = text_dict["text_token_mask"].cpu().numpy()
    for b in which:
        # N4: same host RNG stream -> the same pixels; coordinates within 1e-5 (test_gpu_ingest's bar)
        ing = res.ingested
        assert np.array_equal(ing.sel[b], refs[b]["sel"]) and np.array_equal(ing.view_counts[b], refs[b]["view_counts"])
        got_pts = ing.points[b].cpu().numpy()
        assert_close(got_pts, refs[b]["points"], atol=1e-5, what=f"ingested points, scene {b}")
        # the neck on the GPU's own ingested cloud (one scene per oracle call: eval-mode scenes are independent)
        img = feats[-1][b:b + 1].cpu().numpy()
        ref = oracle.forward(sd, **oracle_kwargs(cfg), points=got_pts[None], text_feats=text[b:b + 1], text_mask=mask[b:b + 1],
                             img_feat=img, num_threads=8)
        m._centers_override = torch.from_numpy(ref["centers"])
        try:
            dbg = m.forward_debug([ing.points[b]], {"text_feats": text_dict["text_feats"][b:b + 1],
                                                    "text_token_mask": text_dict["text_token_mask"][b:b + 1]},
                                  feats[-1][b:b + 1], bbox=ing.bbox[b:b + 1])
        finally:
            m._centers_override = None
        for k in INT_KEYS:
            assert np.array_equal(dbg[k].cpu().numpy().astype(np.int64), ref[k]), (b, k)
        out_b = res.points[b].cpu().numpy()
        assert out_b.shape == ref["outputs"][0].shape, (b, out_b.shape, ref["outputs"][0].shape)
        assert_close(out_b, ref["outputs"][0], atol=1e-4, what=f"preshaped points, scene {b}")
        # ... and the chained (un-injected, batched) call agrees with the single-scene injected call (other launch shapes pick other
        # tile paths: same rows, same order, float noise only)
        assert res.points[b].shape == dbg["outputs"][0].shape
        assert_close(out_b, dbg["outputs"][0].cpu().numpy(), atol=2e-5, what=f"chained vs injected forward, scene {b}")
    # N2 on the GPU's outputs: every voxel row bit-identical (all scenes: the scene column and row order span the batch)
    rc, rf, _ = oracle.voxelize([o.cpu().numpy() for o in res.points], 0.01)
    assert np.array_equal(res.coordinates.cpu().numpy(), rc) and np.array_equal(res.features.cpu().numpy(), rf)
    ends = np.cumsum(np.bincount(rc[:, 0], minlength=len(res.points))).tolist()
    assert res.scene_rows == ends
    vs = np.float32(0.01)
    for li, s in enumerate(MINK_RESNET_STRIDES):
        lref = oracle.level_coordinates(rc, len(res.points), s)
        for b in range(len(res.points)):
            assert np.array_equal(res.level_coords[li][b].cpu().numpy(), lref[b]), (li, b)
        for b in which:
            # N3 on the GPU's level points (DET:429-444): nearest sampling, bit-identical to the restatement
            pts = lref[b].astype(np.float32) * vs
            assert np.array_equal(res.level_points[li][b].cpu().numpy(), pts)
            sc = scenes_np[b]
            sf = sc["img_meta"]["scale_factor"]
            want, nvalid = oracle.point_sample(pts, feats[li][b].cpu().numpy(), projection_matrices(sc["depth2img"]),
                                               scale=(sf[0], sf[1]), pad_hw=(480.0, 480.0), ori_w=480.0)
            got = res.points_imgfeats[b][li].cpu().numpy()
            assert got.shape == (len(pts), FPN_LEVELS[li][0])
            assert np.array_equal(got, want), (li, b, float(np.abs(got - want).max()))
            assert (nvalid > 0).mean() > 0.2, "the level points should project into the views they were scanned from"


def _check_properties(cfg, scenes_np, res):
    """Size-independent properties of every stage, all scenes (nothing here runs the oracle's networks)."""
    B = len(scenes_np)
    ext = np.array([7.0, 5.0, 3.0], np.float32)
    lo = [0] + res.scene_rows[:-1]
    coords = res.coordinates.cpu().numpy()
    assert len(np.unique(coords, axis=0)) == len(coords) and np.all(np.diff(coords[:, 0]) >= 0)
    for b in range(B):
        pts = res.ingested.points[b].cpu().numpy()
        # N4: every point is the back-projection of a depth != 0 pixel -> inside the room (0.5 mm of depth quantisation + fp32)
        assert pts.shape == (cfg.N, 3) and (pts > -5e-3).all() and (pts < ext + 5e-3).all()
        assert res.ingested.sel[b].max() < int(res.ingested.view_counts[b].sum())
        # the neck: original order preserved, points only removed or moved; lengths agree with the voxel stage's input
        out = res.points[b].cpu().numpy()
        assert 0 < len(out) <= cfg.N and np.isfinite(out).all()
        # N2: every surviving point falls into exactly one row of its scene; the rows of scene b are [lo, end)
        vox = np.floor(out / np.float32(0.01)).astype(np.int32)
        rows = coords[lo[b]:res.scene_rows[b]]
        assert (rows[:, 0] == b).all()
        assert len(np.unique(vox, axis=0)) == len(rows)
        # levels: each level's coordinates are multiples of its stride, distinct, and cover the level below
        prev = rows[:, 1:]
        for li, s in enumerate(MINK_RESNET_STRIDES):
            lc = res.level_coords[li][b].cpu().numpy()
            assert (lc % s == 0).all() and len(np.unique(lc, axis=0)) == len(lc)
            cover = np.unique(np.floor_divide(prev, s) * s, axis=0)
            assert len(cover) == len(lc) and np.array_equal(cover, np.unique(lc, axis=0))
            prev = lc
            f = res.points_imgfeats[b][li]
            assert f.shape == (len(lc), FPN_LEVELS[li][0]) and bool(torch.isfinite(f).all())
        assert len(res.level_coords[0][b]) > len(res.level_coords[3][b]) > 0


def test_pipeline_reduced_size_vs_the_oracle_chain():
    """V = 8 views of 480 x 640, N = 20 000 points, the shipped neck configuration (gs 12, ddr 0.6, 3 + 3 blocks), two scenes:
    every stage of both scenes against the oracle chain."""
    base = CONFIGS["cfg4_room"]
    cfg = PreshapeConfig("pipe_small", B=2, N=20000, grid_size=base.grid_size, dynamic_drop_radio=base.dynamic_drop_radio,
                         L=base.L, V=8, text_blocks=3, img_blocks=3, extent=base.extent, seed_base=7300)
    m, sd, pipe, scenes_np, scenes, text_dict, feats, res = _run_chain(cfg, 2, 8, 7300)
    _check_stagewise(cfg, m, sd, scenes_np, text_dict, feats, res, 7300, which=[0, 1])
    _check_properties(cfg, scenes_np, res)
    # the channels-last copies of the feature maps on the caller's stream (inside each sampling call) instead of on the side stream
    # beside the ingest: the same bits
    pipe1 = GroundingFeaturePrefix(m, n_points=cfg.N, overlap_feature_layout=False)
    res1 = pipe1(scenes, text_dict, feats, rng=np.random.RandomState(7300))
    assert torch.equal(res1.coordinates, res.coordinates)
    for b in range(2):
        for li in range(4):
            assert torch.equal(res1.points_imgfeats[b][li], res.points_imgfeats[b][li])
    # float32 depth maps in metres (LoadDepthFromFile's output) give the same clouds as the raw uint16 ones
    m2, _, _, _, _, _, _, res2 = _run_chain(cfg, 2, 8, 7300, as_u16=False)
    for a, b in zip(res.ingested.points, res2.ingested.points):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), atol=1e-5, what="u16 vs f32 depth")


def test_pipeline_at_the_shipped_shape():
    """BASELINE configs[3] as written: 50 views of 480 x 640 depth per scene -> 100 000 points -> gs = 12 (691 kept clusters, 519
    picks) with fp32 features of 50 views -> 1 cm voxels -> four levels, SIX scenes (the training batch, CFG:145) in one chained
    call: properties for all six, the oracle chain stage by stage for two of them."""
    cfg = CONFIGS["cfg4_room"]
    assert (cfg.N, cfg.grid_size, cfg.V, cfg.M_keep, cfg.Kd) == (100000, 12, 50, 691, 519)
    m, sd, pipe, scenes_np, scenes, text_dict, feats, res = _run_chain(cfg, 6, 50, 7400)
    assert [tuple(f.shape[1:]) for f in feats] == [(50, 64, 120, 120), (50, 128, 60, 60), (50, 256, 30, 30), (50, 512, 15, 15)]
    _check_properties(cfg, scenes_np, res)
    _check_stagewise(cfg, m, sd, scenes_np, text_dict, feats, res, 7400, which=[0, 5])
    # a second chained call on the same objects gives the same result (workspaces / staging buffers / lanes reused)
    res2 = pipe(scenes, text_dict, feats, rng=np.random.RandomState(7400), time_stages=True)
    assert torch.equal(res2.coordinates, res.coordinates)
    for b in range(6):
        for li in range(4):
            assert torch.equal(res2.points_imgfeats[b][li], res.points_imgfeats[b][li])
    assert set(res2.stage_ms) == {"ingest", "preshape", "quantize", "levels", "point_sample", "total"}


def test_pipeline_with_the_training_pipelines_3d_augmentation():
    """The TRAIN pipeline (CFG:105-124) adds GlobalRotScaleTrans behind the aggregation: the ingest applies it to the cloud
    (``aug``), and the image-feature sampling has to undo it before projecting (``img_meta['transformation_3d_flow']`` = R, S, T;
    apply_3d_transformation(reverse=True), point_fusion.py:20-107) -- in the chained call the per-scene reverse flows travel with
    the projection matrices through one pinned staging buffer.  Ingest against the oracle (same draws, <= 1e-5); sampled features
    against the oracle's step-by-step reverse flow + point_sample: the composed (3,4) affine of the product differs from the
    step-by-step fp32 flow by ~1e-6 m, which moves a handful of points across a nearest-pixel boundary -- rows that agree must
    agree to 1e-5, and at most 1 % may differ."""
    from oracle import oracle
    base = CONFIGS["cfg4_room"]
    cfg = PreshapeConfig("pipe_aug", B=2, N=20000, grid_size=base.grid_size, dynamic_drop_radio=base.dynamic_drop_radio,
                         L=base.L, V=6, text_blocks=3, img_blocks=3, extent=base.extent, seed_base=7500)
    m, sd = build_module(cfg)
    m = m.cuda()
    scenes_np, scenes, text_dict, feats = _inputs(cfg, 2, 6, 7500)
    for b, (sn, sc) in enumerate(zip(scenes_np, scenes)):
        a = 0.06 * (b + 1)
        rot_T = np.array([[np.cos(a), np.sin(a), 0], [-np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
        aug = dict(rot_mat_T=rot_T, scale=np.float32(1.05 - 0.1 * b), trans=np.array([0.1, -0.05, 0.08], np.float32))
        meta = dict(sn["img_meta"], transformation_3d_flow=["R", "S", "T"], pcd_rotation=rot_T, pcd_scale_factor=float(aug["scale"]),
                    pcd_trans=aug["trans"])
        for d in (sn, sc):
            d["aug"], d["img_meta"] = aug, meta
    pipe = GroundingFeaturePrefix(m, n_points=cfg.N)
    pipe(scenes, text_dict, feats, rng=np.random.RandomState(1))
    torch.cuda.synchronize()
    with _count_synchronises() as calls:
        res = pipe(scenes, text_dict, feats, rng=np.random.RandomState(1))
    assert calls == [], calls
    torch.cuda.synchronize()
    rs = np.random.RandomState(1)
    for b, sc in enumerate(scenes_np):
        depth = sc["depth_img"].astype(np.float32) / np.float32(sc["depth_shift"])
        ref = oracle.ingest(depth, sc["depth_cam2img"], sc["extrinsic"], cfg.N, rng=rs, aug=sc["aug"])
        assert np.array_equal(res.ingested.sel[b], ref["sel"])
        assert_close(res.ingested.points[b].cpu().numpy(), ref["points"], atol=1e-5, what=f"augmented cloud, scene {b}")
        sf = sc["img_meta"]["scale_factor"]
        for li in range(4):
            lp = res.level_points[li][b].cpu().numpy()
            back = oracle.reverse_3d_points(lp, sc["img_meta"], "DEPTH")
            want, nvalid = oracle.point_sample(back, feats[li][b].cpu().numpy(), projection_matrices(sc["depth2img"]),
                                               scale=(sf[0], sf[1]), pad_hw=(480.0, 480.0), ori_w=480.0)
            got = res.points_imgfeats[b][li].cpu().numpy()
            row_err = np.abs(got - want).max(axis=1) if len(lp) else np.zeros(0)
            differ = row_err > 1e-5
            assert differ.mean() <= 0.01, (b, li, float(differ.mean()))
            assert (nvalid > 0).mean() > 0.2
