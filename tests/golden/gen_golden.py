#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the reference itself.

Runs ONLY in the build container (needs /root/reference); the .npz files it
writes are data (inputs + expected outputs) and are committed, this script is
committed with them, and nothing under tests/ reads /root/reference at test time.

How the reference is run:
  * preshape_norm_reverse_drop.py (PRE) is loaded by path, untouched;
  * its three missing imports are replaced by stand-ins registered in sys.modules:
      - embodiedscan.registry.MODELS: register_module()/build() only;
      - timm.models.layers: Mlp (fc1-act-drop1-fc2-drop2), DropPath, trunc_normal_;
      - pytorch3d.ops.sample_farthest_points -> PRE's OWN in-file statement
        ``sample_farthest_points_naive`` (PRE:527-625, "Same Args/Returns");
      - pytorch3d.ops.ball_query -> a torch restatement of the published
        pytorch3d CPU algorithm (first K in index order with dist2 < r*r, idx pad -1),
        written with separate fp32 mul/add tensor ops (no FMA) and gathering through
        PRE's own ``masked_gather`` (PRE:627-672).  pytorch3d is not vendored in the
        reference, so this op's arithmetic is "parity unpinned" (see DESIGN.md).
  * model.eval(), torch.set_num_threads(1) (index_put_ with duplicate targets is only
    deterministic single-threaded), fp32.
  * ``torch.argsort`` inside PRE is unstable and ties are the norm: each case
    is run twice -- as shipped (the permutation torch 2.10 CPU happens to return is
    recorded) and with argsort forced stable at harness level (PRE untouched).  The
    stable run is the pinned production semantics.

Usage:  python tests/golden/gen_golden.py
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference/embodiedscan/models/necks/preshape_norm_reverse_drop.py"

from proxytransformation_amd.synth import PreshapeConfig, fill_state_dict, make_scene_batch  # noqa: E402

CAPTURE = {}


# ----------------------------------------------------------------------------- stand-ins
class _Registry:
    def __init__(self):
        self.d = {}

    def register_module(self):
        def deco(cls):
            self.d[cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.d[cfg.pop("type")](**cfg)


class Mlp(nn.Module):          # timm.models.layers.Mlp surface used at PRE:271
    def __init__(self, in_features, hidden_features=None, out_features=None,
                 act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class DropPath(nn.Module):     # identity in eval mode
    def __init__(self, p=0.):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0.:
            return x
        keep = 1 - self.p
        m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * m / keep


def trunc_normal_(t, std=1.):
    return nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std)


def _install_standins():
    reg = _Registry()
    es = types.ModuleType("embodiedscan")
    esr = types.ModuleType("embodiedscan.registry")
    esr.MODELS = reg
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tml = types.ModuleType("timm.models.layers")
    tml.Mlp, tml.DropPath, tml.trunc_normal_ = Mlp, DropPath, trunc_normal_
    p3 = types.ModuleType("pytorch3d")
    p3o = types.ModuleType("pytorch3d.ops")
    p3o.ball_query = lambda *a, **k: _ball_query(*a, **k)
    p3o.sample_farthest_points = lambda *a, **k: _fps(*a, **k)
    sys.modules.update({"embodiedscan": es, "embodiedscan.registry": esr, "timm": timm,
                        "timm.models": tm, "timm.models.layers": tml,
                        "pytorch3d": p3, "pytorch3d.ops": p3o})
    return reg


PRE = None


def _ball_query(p1, p2, lengths1=None, lengths2=None, K=500, radius=0.2, return_nn=True):
    """torch restatement of pytorch3d's ball_query_cpu.cpp (see module docstring)."""
    B, M, _ = p1.shape
    N = p2.shape[1]
    r2 = torch.tensor(float(radius), dtype=torch.float32) * torch.tensor(float(radius), dtype=torch.float32)
    idx = torch.full((B, M, K), -1, dtype=torch.int64)
    dists = torch.zeros((B, M, K), dtype=p1.dtype)
    chunk = 64
    for b in range(B):
        for m0 in range(0, M, chunk):
            c = p1[b, m0:m0 + chunk]                                   # (c,3)
            d0 = c[:, None, 0] - p2[b, None, :, 0]
            d1 = c[:, None, 1] - p2[b, None, :, 1]
            d2_ = c[:, None, 2] - p2[b, None, :, 2]
            acc = d0 * d0                                              # 0 + dx*dx == dx*dx
            acc = acc + d1 * d1
            acc = acc + d2_ * d2_
            hit = acc < r2
            rank = hit.cumsum(dim=1) - 1                                # slot of each hit
            sel = hit & (rank < K)
            mm, jj = sel.nonzero(as_tuple=True)
            idx[b, m0 + mm, rank[mm, jj]] = jj
            dists[b, m0 + mm, rank[mm, jj]] = acc[mm, jj]
    nn_ = PRE.masked_gather(p2, idx)
    CAPTURE.setdefault("bq", []).append((p1.clone(), idx.clone(), nn_.clone()))
    return dists, idx, nn_


def _fps(points, lengths=None, K=50, random_start_point=False):
    pts, idx = PRE.sample_farthest_points_naive(points, lengths, K, random_start_point)
    CAPTURE["fps_in"] = points.clone()
    CAPTURE["fps_idx"] = idx.clone()
    return pts, idx


# ----------------------------------------------------------------------------- cases
CASES = [
    # cfg1 shape (BASELINE.json configs[0]) with B=2 so the masked scene (index 1) exists
    PreshapeConfig("g1_cfg1", B=2, N=20000, grid_size=8, dynamic_drop_radio=0.875,
                   L=16, V=4, seed_base=1000),
    # sparse scene: ~26 points per ball -> padded slots, padding-count ordering matters,
    # 2+2 blocks so "only the last block is live"  is exercised
    PreshapeConfig("g2_sparse", B=2, N=300, grid_size=4, dynamic_drop_radio=0.5,
                   L=8, V=2, text_blocks=2, img_blocks=2, seed_base=7100),
    # small room: every side < 2*margin -> the grid inverts and centres are clamped onto
    # the bbox; many coincident centres, FPS ties
    PreshapeConfig("g3_room", B=1, N=5000, grid_size=4, dynamic_drop_radio=0.5,
                   L=5, V=3, extent=(7.0, 5.0, 3.0), seed_base=7300),
]


def run_case(reg, cfg: PreshapeConfig):
    torch.manual_seed(0)
    model = reg.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
    sd_np = fill_state_dict(model.state_dict())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model.eval()
    pts, text, mask, img = make_scene_batch(cfg)
    tpts = [torch.from_numpy(pts[b].copy()) for b in range(cfg.B)]
    text_dict = {"text_feats": torch.from_numpy(text), "text_token_mask": torch.from_numpy(mask)}
    timg = torch.from_numpy(img)

    hooks = {}

    def hook(name):
        def fn(mod, inp, outp):
            hooks[name] = outp
        return fn

    hs = [model.get_deformable_cluster.get_offsets.register_forward_hook(hook("offset_raw")),
          model.simple_encoder.register_forward_hook(hook("point_proxy")),
          model.norm_img.register_forward_hook(hook("img_proxy")),
          model.text_norm[-1].register_forward_hook(hook("text_guide")),
          model.img_norm[-1].register_forward_hook(hook("img_guide")),
          model.text_trans_norm.register_forward_hook(hook("translate_t")),
          model.img_trans_norm.register_forward_hook(hook("transform_t")),
          model.textformer[-1].register_forward_hook(hook("text_block")),
          model.imgformer[-1].register_forward_hook(hook("img_block"))]

    # dynamic_cluster_dropout's own results are not visible through module hooks
    orig_drop = model.dynamic_cluster_dropout
    drop_out = {}

    def wrapped_drop(cluster, center, idx, empty_drop=0.3):
        r = orig_drop(cluster, center, idx, empty_drop)
        drop_out.update(new_cluster=r[0], new_center=r[1], new_idx=r[2], drop_idx=r[3])
        return r
    model.dynamic_cluster_dropout = wrapped_drop

    orig_replace = PRE.pt_replace
    replaced = {}

    def wrapped_replace(p2, idx, cluster):
        replaced["new_clusters"] = cluster.clone()
        r = orig_replace(p2, idx, cluster)
        replaced["new_points"] = r.clone()
        return r
    PRE.pt_replace = wrapped_replace

    real_argsort = torch.argsort
    results = {}
    for mode in ("shipped", "stable"):
        CAPTURE.clear()
        captured_sorted = {}

        def argsort(x, dim=-1, descending=False, stable=False):
            r = real_argsort(x, dim=dim, descending=descending, stable=(mode == "stable"))
            captured_sorted["sorted"] = r.clone()
            captured_sorted["pad_counts"] = x.clone()
            return r
        torch.argsort = argsort
        try:
            torch.set_num_threads(1)
            with torch.no_grad():
                outs = model(tpts, text_dict, timg)
        finally:
            torch.argsort = real_argsort
        results[mode] = dict(
            outs=[o.numpy().copy() for o in outs],
            sorted=captured_sorted["sorted"].numpy().copy(),
            pad_counts=captured_sorted["pad_counts"].numpy().copy(),
            bq=[(a.numpy().copy(), b.numpy().copy(), c.numpy().copy()) for a, b, c in CAPTURE["bq"]],
            fps_idx=CAPTURE["fps_idx"].numpy().copy(),
            hooks={k: (v.detach().numpy().copy()) for k, v in hooks.items()},
            drop={k: v.numpy().copy() for k, v in drop_out.items()},
            replaced={k: v.numpy().copy() for k, v in replaced.items()},
        )
    PRE.pt_replace = orig_replace
    for h in hs:
        h.remove()

    st, sh = results["stable"], results["shipped"]
    Mt = cfg.Mt
    save = dict(
        # ---- inputs
        points=pts, text_feats=text, text_mask=mask, img_feat=img,
        cfg=np.array([cfg.B, cfg.N, cfg.grid_size, cfg.L, cfg.V, cfg.embed_dim, cfg.num_heads,
                      cfg.num_sub, cfg.text_blocks, cfg.img_blocks], np.int64),
        dynamic_drop_radio=np.float64(cfg.dynamic_drop_radio),
        extent=np.asarray(cfg.extent, np.float64), seed_base=np.int64(cfg.seed_base),
        # ---- clustering (identical in both modes up to the argsort)
        centers0=st["bq"][0][0], cluster1=st["bq"][0][2],
        offset_raw=st["hooks"]["offset_raw"],
        centers=st["bq"][1][0], idx2=st["bq"][1][1], cluster2=st["bq"][1][2],
        pad_counts=st["pad_counts"],
        # ---- stable (production pin)
        order_stable=st["sorted"][:, :Mt], fps_stable=st["fps_idx"],
        kcluster_stable=st["drop"]["new_cluster"], kcenter_stable=st["drop"]["new_center"],
        kidx_stable=st["drop"]["new_idx"], drop_idx_stable=st["drop"]["drop_idx"],
        point_proxy_stable=st["hooks"]["point_proxy"], img_proxy=st["hooks"]["img_proxy"],
        text_guide_stable=st["hooks"]["text_guide"], img_guide_stable=st["hooks"]["img_guide"],
        text_block_stable=st["hooks"]["text_block"], img_block_stable=st["hooks"]["img_block"],
        translate_stable=st["hooks"]["translate_t"].transpose(0, 2, 1),
        transform_stable=st["hooks"]["transform_t"].transpose(0, 2, 1),
        new_clusters_stable=st["replaced"]["new_clusters"],
        new_points_stable=st["replaced"]["new_points"],
        # ---- as shipped (torch 2.10 CPU unstable argsort), replayed through order_override
        order_shipped=sh["sorted"][:, :Mt], fps_shipped=sh["fps_idx"],
        kidx_shipped=sh["drop"]["new_idx"], drop_idx_shipped=sh["drop"]["drop_idx"],
        translate_shipped=sh["hooks"]["translate_t"].transpose(0, 2, 1),
        transform_shipped=sh["hooks"]["transform_t"].transpose(0, 2, 1),
        new_points_shipped=sh["replaced"]["new_points"],
    )
    for b in range(cfg.B):
        save[f"out_stable_{b}"] = st["outs"][b]
        save[f"out_shipped_{b}"] = sh["outs"][b]
    # boundary-safety report: smallest |dist2 - r^2| over scanned candidates of
    # ball query #2, so end-to-end GPU tests know whether membership may legitimately flip
    c2, i2 = st["bq"][1][0], st["bq"][1][1]
    margins = []
    for b in range(cfg.B):
        best = np.inf
        for m in range(c2.shape[1]):
            last = i2[b, m].max()
            stop = (last + 1) if (i2[b, m] >= 0).all() else cfg.N
            d = c2[b, m][None, :] - pts[b, :stop]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            best = min(best, float(np.abs(d2 - np.float32(9.0)).min()))
        margins.append(best)
    save["bq2_boundary_margin"] = np.asarray(margins, np.float64)
    path = os.path.join(HERE, cfg.name + ".npz")
    np.savez_compressed(path, **save)
    n_par = sum(int(np.prod(v.shape)) for k, v in sd_np.items())
    print(f"{cfg.name}: params={n_par} outs={[o.shape for o in st['outs']]} "
          f"order_equal={np.array_equal(st['sorted'], sh['sorted'])} "
          f"margin={margins} -> {os.path.getsize(path) / 1e6:.2f} MB")


# ----------------------------------------------------------------------------- train mode
TRAIN_CASE = PreshapeConfig("g4_train", B=2, N=3000, grid_size=4, dynamic_drop_radio=0.5, L=6, V=3,
                            text_blocks=2, img_blocks=2, seed_base=7500)
GRAD_SAMPLES = 4096


def loss_weights(b: int, n: int) -> np.ndarray:
    """Fixed upstream gradient of output b: d loss / d out_b[i, d] (the tests rebuild it from the same formula)."""
    i = np.arange(n, dtype=np.float64)[:, None]
    d = np.arange(3, dtype=np.float64)[None, :]
    return np.sin(0.37 * i + 1.3 * d + 0.7 * b).astype(np.float32)


def sample_idx(numel: int) -> np.ndarray:
    step = max(1, -(-numel // GRAD_SAMPLES))
    return np.arange(0, numel, step, dtype=np.int64)


def run_train_case(reg, cfg: PreshapeConfig):
    """One optimisation-free training step of the reference: model.train() (batch-statistics BatchNorm, running-stat
    update), all drop rates 0 (deterministic), loss = sum_b <out_b, W_b>, backward.  Captures the train-mode
    intermediates, the gradient of every parameter and input (sampled for big tensors) and which parameters get none."""
    torch.manual_seed(0)
    model = reg.build(dict(type="ProxyTransformationNormReverse", drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
                           **cfg.module_kwargs()))
    sd_np = fill_state_dict(model.state_dict())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model.train()
    pts, text, mask, img = make_scene_batch(cfg)
    tpts = [torch.from_numpy(pts[b].copy()) for b in range(cfg.B)]
    ttext = torch.from_numpy(text).requires_grad_(True)
    timg = torch.from_numpy(img).requires_grad_(True)
    text_dict = {"text_feats": ttext, "text_token_mask": torch.from_numpy(mask)}
    hooks = {}

    def hook(name):
        def fn(mod, inp, outp):
            hooks[name] = outp.detach().clone()
        return fn
    hs = [model.simple_encoder.register_forward_hook(hook("point_proxy")),
          model.norm_img.register_forward_hook(hook("img_proxy")),
          model.text_trans_norm.register_forward_hook(hook("translate_t")),
          model.img_trans_norm.register_forward_hook(hook("transform_t"))]
    real_argsort = torch.argsort
    captured = {}

    def argsort(x, dim=-1, descending=False, stable=False):
        r = real_argsort(x, dim=dim, descending=descending, stable=True)
        captured["sorted"] = r.clone()
        return r
    CAPTURE.clear()
    torch.argsort = argsort
    try:
        torch.set_num_threads(1)
        outs = model(tpts, text_dict, timg)
        loss = sum((o * torch.from_numpy(loss_weights(b, o.shape[0]))).sum() for b, o in enumerate(outs))
        loss.backward()
    finally:
        torch.argsort = real_argsort
    for h in hs:
        h.remove()
    save = dict(points=pts, text_feats=text, text_mask=mask, img_feat=img,
                cfg=np.array([cfg.B, cfg.N, cfg.grid_size, cfg.L, cfg.V, cfg.embed_dim, cfg.num_heads, cfg.num_sub,
                              cfg.text_blocks, cfg.img_blocks], np.int64),
                dynamic_drop_radio=np.float64(cfg.dynamic_drop_radio), extent=np.asarray(cfg.extent, np.float64),
                seed_base=np.int64(cfg.seed_base), loss=np.float64(loss.item()),
                centers=CAPTURE["bq"][1][0].detach().numpy().copy(), idx2=CAPTURE["bq"][1][1].numpy().copy(),
                order=captured["sorted"][:, :cfg.Mt].numpy().copy(), fps=CAPTURE["fps_idx"].numpy().copy(),
                point_proxy=hooks["point_proxy"].numpy(), img_proxy=hooks["img_proxy"].numpy(),
                translate=hooks["translate_t"].numpy().transpose(0, 2, 1),
                transform=hooks["transform_t"].numpy().transpose(0, 2, 1))
    for b, o in enumerate(outs):
        save[f"out_{b}"] = o.detach().numpy().copy()
    none = []
    for name, prm in list(model.named_parameters()) + [("input.text_feats", ttext), ("input.img_feat", timg)]:
        if prm.grad is None:
            none.append(name)
            continue
        g = prm.grad.detach().numpy().reshape(-1)
        idx = sample_idx(g.size)
        save["grad." + name] = g[idx].copy()
        save["gnorm." + name] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        save["gsum." + name] = np.float64(g.astype(np.float64).sum())
    save["none_grads"] = np.array(none)
    for name, buf in model.named_buffers():                       # running statistics after the step
        save["buf." + name] = buf.detach().numpy().copy()
    path = os.path.join(HERE, cfg.name + ".npz")
    np.savez_compressed(path, **save)
    print(f"{cfg.name}: loss={loss.item():.6f} outs={[tuple(o.shape) for o in outs]} none_grads={len(none)} "
          f"-> {os.path.getsize(path) / 1e6:.2f} MB")


# ----------------------------------------------------------------------------- image -> point sampling
def gen_point_sample():
    """batch_point_sample (models/layers/fusion_layers/point_fusion.py:208-313) exactly as the detector calls it
    (detectors/sparse_featfusion_grounder_preshape.py:428-444: aligned=False -> nearest, zeros padding, align_corners,
    valid_flag), run from the reference file with stand-ins for its framework imports.  Points whose projection lands
    within 1e-3 px of a rounding / validity boundary in any view are filtered out (fp32 bmm order is not pinned)."""
    import types
    ref = "/root/reference/embodiedscan"
    mm = types.ModuleType("mmcv"); mmc = types.ModuleType("mmcv.cnn"); mmc.ConvModule = nn.Module
    me = types.ModuleType("mmengine"); mem = types.ModuleType("mmengine.model"); mem.BaseModule = nn.Module
    p3t = types.ModuleType("pytorch3d.transforms"); p3t.euler_angles_to_matrix = lambda *a, **k: None
    eu = types.ModuleType("embodiedscan.utils"); eu.ConfigType = dict
    sys.modules.update({"mmcv": mm, "mmcv.cnn": mmc, "mmengine": me, "mmengine.model": mem, "pytorch3d.transforms": p3t,
                        "embodiedscan.utils": eu})

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    load("embodiedscan.utils.array_converter", ref + "/utils/array_converter.py")
    b3 = load("embodiedscan.structures.bbox_3d", ref + "/structures/bbox_3d/utils.py")
    b3.get_proj_mat_by_coord_type = lambda *a, **k: None
    # the reference's own point containers (DepthPoints.flip / rotate / scale / translate run the reverse 3D flow)
    es_st = types.ModuleType("embodiedscan.structures"); es_st.__path__ = []
    sys.modules.setdefault("embodiedscan.structures", es_st)
    sys.modules["embodiedscan.structures.bbox_3d.utils"] = b3
    spec = importlib.util.spec_from_file_location("embodiedscan.structures.points", ref + "/structures/points/__init__.py",
                                                  submodule_search_locations=[ref + "/structures/points"])
    pts_mod = importlib.util.module_from_spec(spec)
    sys.modules["embodiedscan.structures.points"] = pts_mod
    spec.loader.exec_module(pts_mod)
    pf = load("pf_ref", ref + "/models/layers/fusion_layers/point_fusion.py")

    rng = np.random.default_rng(123)
    V, C, H, W, N = 6, 32, 30, 40, 4000
    pad_h, pad_w = 480.0, 640.0
    feats = rng.standard_normal((V, C, H, W), dtype=np.float32)
    proj = np.zeros((V, 4, 4), np.float32)
    for v in range(V):                                   # pin-hole cameras on a ring looking at the scene centre
        ang = 2 * np.pi * v / V
        R = np.array([[np.cos(ang), 0, -np.sin(ang)], [0, 1, 0], [np.sin(ang), 0, np.cos(ang)]], np.float64)
        t = np.array([0.3 * v - 0.5, 0.2, 4.0 + 0.3 * v])
        ext = np.eye(4); ext[:3, :3] = R; ext[:3, 3] = t
        K = np.eye(4); K[0, 0] = K[1, 1] = 420.0 + 10 * v; K[0, 2] = 320.0; K[1, 2] = 240.0
        proj[v] = (K @ ext).astype(np.float32)
    points = ((rng.random((N, 3)) - 0.5) * np.array([9.0, 6.0, 9.0])).astype(np.float32)
    save = dict(feats=feats, proj=proj, pad=np.array([pad_h, pad_w], np.float32))
    # 3D augmentation as the training pipeline records it (GlobalRotScaleTrans: 'R', 'S', 'T'; RandomFlip3D: 'HF'):
    # the sampled points live in the AUGMENTED frame and are taken back by apply_3d_transformation(reverse=True)
    ang = 0.07
    rot_T = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]], np.float32)
    meta3d = dict(transformation_3d_flow=["HF", "R", "S", "T"], pcd_horizontal_flip=True, pcd_vertical_flip=False,
                  pcd_rotation=rot_T, pcd_scale_factor=1.07, pcd_trans=np.array([0.11, -0.06, 0.04], np.float32))
    fwd = points.astype(np.float64).copy()
    fwd[:, 0] = -fwd[:, 0]
    fwd = (fwd @ rot_T.astype(np.float64)) * 1.07 + meta3d["pcd_trans"].astype(np.float64)
    points_aug3d = fwd.astype(np.float32)
    save.update(flow3d_rot_T=rot_T, flow3d_scale=np.float32(1.07), flow3d_trans=meta3d["pcd_trans"])
    for name, scale, crop, flip, meta, aligned in (("plain", (1.0, 1.0), (0.0, 0.0), False, {}, False),
                                                   ("aug", (0.9, 1.1), (12.0, -7.0), True, {}, False),
                                                   ("flow3d", (0.9, 1.1), (12.0, -7.0), False, meta3d, False),
                                                   ("bilinear", (1.0, 1.0), (3.0, 2.0), False, {}, True)):
        src_points = points_aug3d if meta else points
        pts_t, proj_t = torch.from_numpy(src_points), torch.from_numpy(proj)
        # boundary filter, in float64 (on the un-augmented positions the projection sees)
        p4 = np.concatenate([points.astype(np.float64), np.ones((N, 1))], 1)
        q = np.einsum("vrk,nk->vnr", proj.astype(np.float64), p4)
        z = np.maximum(q[..., 2], 1e-3)
        cx = q[..., 0] / z * scale[0] - crop[0]
        cy = q[..., 1] / z * scale[1] - crop[1]
        if flip:
            cx = 480.0 * 1.3 - cx
        ix = ((cx / pad_w * 2 - 1) + 1) / 2 * (W - 1)
        iy = ((cy / pad_h * 2 - 1) + 1) / 2 * (H - 1)
        safe = np.ones(N, bool)
        for arr, lim in ((ix, None), (iy, None)):
            frac = np.abs(arr - np.floor(arr) - 0.5) if not aligned else np.minimum(arr - np.floor(arr), np.ceil(arr) - arr)
            safe &= (frac > (2e-3 if meta or aligned else 1e-3)).all(0)      # nearest: away from x.5; bilinear: from integers
        for arr, hi in ((cx, pad_w), (cy, pad_h)):
            safe &= (np.abs(arr) > 1e-2).all(0) & (np.abs(arr - hi) > 1e-2).all(0)
        safe &= (np.abs(q[..., 2]) > 1e-2).all(0) & (np.abs(q[..., 2] - 1e-3) > 1e-4).all(0)
        sel = np.nonzero(safe)[0][:1500]
        out = pf.batch_point_sample(dict(meta), img_features=torch.from_numpy(feats), points=pts_t[sel], proj_mat=proj_t,
                                    coord_type="DEPTH", img_scale_factor=torch.tensor(scale), img_crop_offset=torch.tensor(crop),
                                    img_flip=flip, img_pad_shape=(int(pad_h), int(pad_w)), img_shape=(600, int(480 * 1.3)),
                                    aligned=aligned)
        save[f"{name}_points"] = src_points[sel]
        save[f"{name}_out"] = out.numpy()
        save[f"{name}_cfg"] = np.array([scale[0], scale[1], crop[0], crop[1], float(flip), 480 * 1.3], np.float32)
        print(f"g5_point_sample/{name}: {len(sel)} points, nonzero rows {(np.abs(out.numpy()).sum(1) > 0).sum()}")
    # the reverse 3D flow alone, per coordinate type and flip (the flips negate different axes: depth_points.py:47-50,
    # lidar_points.py:47-50, cam_points.py:47-50): apply_3d_transformation(reverse=True) itself
    save["rev3d_points"] = points_aug3d[:256]
    for ct in ("DEPTH", "LIDAR", "CAMERA"):
        for fl in ("hf", "vf"):
            meta_f = dict(transformation_3d_flow=["VF", "HF", "R", "S", "T"], pcd_horizontal_flip=fl == "hf",
                          pcd_vertical_flip=fl == "vf", pcd_rotation=rot_T, pcd_scale_factor=1.07, pcd_trans=meta3d["pcd_trans"])
            save[f"rev3d_{ct}_{fl}"] = pf.apply_3d_transformation(torch.from_numpy(points_aug3d[:256]), ct, meta_f,
                                                                  reverse=True).numpy()
    path = os.path.join(HERE, "g5_point_sample.npz")
    np.savez_compressed(path, **save)
    print(f"g5_point_sample -> {os.path.getsize(path) / 1e6:.2f} MB")


# ----------------------------------------------------------------------------- multi-view depth ingest
def gen_ingest():
    """The reference's own transform classes -- MultiViewPipeline(ConvertRGBDToPoints, PointSample) ->
    AggregateMultiViewPoints -> PointSample (-> GlobalRotScaleTrans), as configured at CFG:105-142 -- run from the
    reference files on small synthetic depth maps with ``np.random.seed`` fixed.  Stand-ins only for the framework
    imports (mmcv BaseTransform / Compose / imresize, the TRANSFORMS registry, mmdet's RandomFlip base class) and for
    the file loader (the depth maps are handed over in memory)."""
    import types
    ref = "/root/reference/embodiedscan"

    class BaseTransform:
        def __call__(self, results):
            return self.transform(results)

    class Compose:
        def __init__(self, transforms):
            self.transforms = list(transforms)

        def __call__(self, data):
            for t in self.transforms:
                data = t(data)
            return data

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls
    mm = types.ModuleType("mmcv"); mm.imresize = None; mm.__path__ = []
    mmt = types.ModuleType("mmcv.transforms"); mmt.BaseTransform = BaseTransform; mmt.Compose = Compose
    md = types.ModuleType("mmdet"); md.__path__ = []
    mdd = types.ModuleType("mmdet.datasets"); mdd.__path__ = []
    mddt = types.ModuleType("mmdet.datasets.transforms"); mddt.RandomFlip = BaseTransform
    es = types.ModuleType("embodiedscan"); es.__path__ = []
    esr = types.ModuleType("embodiedscan.registry"); esr.TRANSFORMS = _Reg()
    est = types.ModuleType("embodiedscan.structures"); est.__path__ = []
    eu = types.ModuleType("embodiedscan.utils"); eu.__path__ = []
    p3 = types.ModuleType("pytorch3d"); p3.__path__ = []
    p3t = types.ModuleType("pytorch3d.transforms"); p3t.euler_angles_to_matrix = lambda *a, **k: None
    sys.modules.update({"pytorch3d": p3, "pytorch3d.transforms": p3t, "mmcv": mm, "mmcv.transforms": mmt, "mmdet": md, "mmdet.datasets": mdd,
                        "mmdet.datasets.transforms": mddt, "embodiedscan": es, "embodiedscan.registry": esr,
                        "embodiedscan.structures": est, "embodiedscan.utils": eu})

    def load(name, path, package=False):
        spec = importlib.util.spec_from_file_location(
            name, path, submodule_search_locations=[os.path.dirname(path)] if package else None)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    load("embodiedscan.utils.array_converter", ref + "/utils/array_converter.py")
    b3 = types.ModuleType("embodiedscan.structures.bbox_3d"); b3.__path__ = []
    sys.modules["embodiedscan.structures.bbox_3d"] = b3
    b3u = load("embodiedscan.structures.bbox_3d.utils", ref + "/structures/bbox_3d/utils.py")
    b3.points_cam2img, b3.points_img2cam = b3u.points_cam2img, b3u.points_img2cam
    load("embodiedscan.structures.points", ref + "/structures/points/__init__.py", package=True)
    tp = load("ref_points_tf", ref + "/datasets/transforms/points.py")
    tm = load("ref_multiview_tf", ref + "/datasets/transforms/multiview.py")
    ta = load("ref_augmentation_tf", ref + "/datasets/transforms/augmentation.py")

    rng = np.random.default_rng(606)
    V, H, W, n_points = 5, 36, 48, 3000
    depth = (1.0 + 3.5 * rng.random((V, H, W))).astype(np.float32)
    depth[rng.random((V, H, W)) < 0.25] = 0.0                    # holes, like a real sensor
    depth[2, :, : W // 2] = 0.0
    depth[3, 5:, :] = 0.0                                         # few points: sampled WITH replacement (points.py:396)
    raw = np.round(depth * 1000.0).astype(np.uint16)              # the decoded 16-bit image; depth_shift = 1000
    depth = raw.astype(np.float32) / 1000.0                       # LoadDepthFromFile (loading.py:135-136)
    K = np.array([[40.0, 0.0, 23.5, 0.0], [0.0, 41.0, 17.5, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    ext = []
    for v in range(V):                                           # global2ego of cameras on a ring (mv_3dvg_dataset.py:547: float32)
        ang = 2 * np.pi * v / V
        R = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]]) @ \
            np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])
        cam2global = np.eye(4); cam2global[:3, :3] = R; cam2global[:3, 3] = [1.5 * np.cos(ang), 1.5 * np.sin(ang), 1.2 + 0.1 * v]
        ext.append(np.linalg.inv(cam2global).astype(np.float32))

    def load_depth(results):                                     # in-memory LoadDepthFromFile
        results["depth_img"] = depth[results["depth_img_path"]]
        return results
    save = dict(depth_u16=raw, depth_shift=np.float32(1000.0), depth_cam2img=K, extrinsic=np.stack(ext),
                n_points=np.int64(n_points))
    for name, with_aug, seed in (("plain", False, 11), ("aug", True, 12)):
        np.random.seed(seed)
        results = dict(img_path=list(range(V)), depth_img_path=list(range(V)), depth_cam2img=K.copy(), cam2img=K.copy(),
                       depth_shift=1000.0, depth2img=dict(intrinsic=K.copy(), extrinsic=[e.copy() for e in ext]))
        pipe = [tm.MultiViewPipeline(n_images=V, ordered=True,
                                     transforms=[load_depth, tp.ConvertRGBDToPoints(coord_type="CAMERA"),
                                                 tp.PointSample(num_points=n_points // 10)]),
                tm.AggregateMultiViewPoints(coord_type="DEPTH"), tp.PointSample(num_points=n_points)]
        if with_aug:
            pipe.append(ta.GlobalRotScaleTrans(rot_range=[-0.087266, 0.087266], scale_ratio_range=[.9, 1.1],
                                               translation_std=[.1, .1, .1], shift_height=False))
        for t in pipe:
            results = t(results)
        pts = results["points"].tensor.numpy()
        assert pts.shape == (n_points, 3) and pts.dtype == np.float32
        save[f"{name}_points"] = pts
        save[f"{name}_seed"] = np.int64(seed)
        if with_aug:
            save["aug_rot_mat_T"] = np.asarray(results["pcd_rotation"], np.float32)
            save["aug_scale"] = np.float32(results["pcd_scale_factor"])
            save["aug_trans"] = np.asarray(results["pcd_trans"], np.float32)
        print(f"g6_ingest/{name}: {pts.shape}, bbox {pts.min(0)} .. {pts.max(0)}")
    path = os.path.join(HERE, "g6_ingest.npz")
    np.savez_compressed(path, **save)
    print(f"g6_ingest -> {os.path.getsize(path) / 1e6:.2f} MB")


def write_manifest(reg):
    """Key / shape / dtype manifest of the reference module's state_dict (data, not code)."""
    import json
    entries = []
    for kwargs in (dict(), dict(n_points=100000, grid_size=12, text_blocks=3, img_blocks=3,
                                dynamic_drop_radio=0.6, num_sub=30),            # CFG:41
                   dict(grid_size=8, dynamic_drop_radio=0.5, qkv_bias=True)):
        model = reg.build(dict(type="ProxyTransformationNormReverse", **kwargs))
        entries.append(dict(kwargs=kwargs, tensors=[[k, list(v.shape), str(v.dtype)]
                                                    for k, v in model.state_dict().items()]))
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(entries, f, indent=0)
    print("manifest:", [len(e["tensors"]) for e in entries], "tensors")


def main():
    global PRE
    reg = _install_standins()
    spec = importlib.util.spec_from_file_location("pre_ref", REF)
    PRE = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(PRE)
    only = sys.argv[1:]
    if not only or "manifest" in only:
        write_manifest(reg)
    for cfg in CASES:
        if only and cfg.name not in only:
            continue
        run_case(reg, cfg)
    if not only or TRAIN_CASE.name in only:
        run_train_case(reg, TRAIN_CASE)
    if not only or "g5_point_sample" in only:
        gen_point_sample()
    if not only or "g6_ingest" in only:
        gen_ingest()


if __name__ == "__main__":
    main()
