"""CPU oracle for the preshape hot path -- TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  ``proxytransformation_amd`` must not
(tests/test_host_cpu.py::test_product_never_touches_the_oracle enforces that).

It restates ``ProxyTransformationNormReverse.forward``
(PRE = embodiedscan/models/necks/preshape_norm_reverse_drop.py:424-469 of the
reference) as a flat function over a ``state_dict``:

* index-producing steps (grid centres, ball query, cluster selection + FPS,
  scatter ownership, point removal) run in plain C: ``oracle/ptx_oracle.c``;
* the floating-point networks (offset net, PointNet, attention pooling, proxy
  blocks, heads) are a torch-CPU fp32 reference written with ``torch.nn.functional``.

Pinning: ``tests/test_oracle_golden.py`` checks every intermediate this module
returns against ``tests/golden/*.npz``, which were captured from the reference
file itself (see tests/golden/gen_golden.py).  pytorch3d's ball-query
arithmetic is third-party and un-vendored: **parity unpinned** for that op
(details in the header of ptx_oracle.c and in DESIGN.md).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RADIUS = 3.0        # PRE:23 (not configurable from the registry config)
MARGIN = 4.0        # PRE:23
EMPTY_DROP = 0.3    # PRE:352
BN_EPS = 1e-5
LN_EPS = 1e-5


def build_lib(force: bool = False) -> str:
    """Compile oracle/ptx_oracle.c with the committed Makefile; returns the .so path."""
    so = os.path.join(_HERE, "libptx_oracle.so")
    src = os.path.join(_HERE, "ptx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libptx_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_lib())
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _i64(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.int64))


# --------------------------------------------------------------------------
# C-backed index steps
# --------------------------------------------------------------------------
def grid_centers(points: np.ndarray, gs: int, margin: float = MARGIN):
    """PRE:33-51.  linspace comes from torch."""
    points = _f32(points)
    B, N, _ = points.shape
    lin = torch.linspace(0, 1, gs).numpy().astype(np.float32)
    M = gs ** 3
    centers = np.empty((B, M, 3), np.float32)
    mn = np.empty((B, 3), np.float32)
    mx = np.empty((B, 3), np.float32)
    lib().oracle_grid_centers(_p(points), B, N, _p(lin), gs, ctypes.c_float(margin),
                              _p(centers), _p(mn), _p(mx))
    return centers, mn, mx


def ball_query(centers: np.ndarray, points: np.ndarray, K: int, radius: float = RADIUS,
               want_scanned: bool = False):
    """pytorch3d.ops.ball_query(p1=centers, p2=points, K, radius) -> idx, gathered xyz."""
    centers, points = _f32(centers), _f32(points)
    B, M, _ = centers.shape
    N = points.shape[1]
    idx = np.empty((B, M, K), np.int64)
    cluster = np.empty((B, M, K, 3), np.float32)
    scanned = np.empty((B, M), np.int32) if want_scanned else None
    lib().oracle_ball_query(_p(centers), _p(points), B, M, N, K, ctypes.c_float(radius),
                            _p(idx), _p(cluster), _p(scanned) if want_scanned else None)
    return (idx, cluster, scanned) if want_scanned else (idx, cluster)


def fps(pts: np.ndarray, Kd: int) -> np.ndarray:
    pts = _f32(pts)
    B, P, _ = pts.shape
    picks = np.empty((B, Kd), np.int64)
    lib().oracle_fps(_p(pts), B, P, Kd, _p(picks))
    return picks


def select_clusters(idx: np.ndarray, centers: np.ndarray, Mt: int, Mk: int,
                    order_override: Optional[np.ndarray] = None):
    """PRE:352-420 -> dict(pad_counts, order, picks, keep) (positions, see ptx_oracle.c)."""
    idx, centers = _i64(idx), _f32(centers)
    B, M, K = idx.shape
    Kd = Mt - Mk
    pad = np.empty((B, M), np.int64)
    order = np.empty((B, Mt), np.int64)
    picks = np.empty((B, Kd), np.int64)
    keep = np.empty((B, Mk), np.int64)
    ov = _i64(order_override) if order_override is not None else None
    lib().oracle_select_clusters(_p(idx), _p(centers), B, M, K, Mt, Mk,
                                 _p(ov) if ov is not None else None,
                                 _p(pad), _p(order), _p(picks), _p(keep))
    return dict(pad_counts=pad, order=order, picks=picks, keep=keep)


def pt_replace(points: np.ndarray, idx: np.ndarray, newc: np.ndarray) -> np.ndarray:
    out = _f32(points).copy()
    idx, newc = _i64(idx), _f32(newc)
    B, N, _ = out.shape
    _, Mk, K = idx.shape
    lib().oracle_pt_replace(_p(out), _p(idx), _p(newc), B, N, Mk, K)
    return out


def remove_points(points: np.ndarray, drop_idx: np.ndarray):
    points, drop_idx = _f32(points), _i64(drop_idx)
    B, N, _ = points.shape
    Nd = drop_idx.shape[1]
    out = np.empty_like(points)
    counts = np.empty((B,), np.int64)
    lib().oracle_remove_points(_p(points), _p(drop_idx), B, N, Nd, _p(out), _p(counts))
    return [out[b, :counts[b]].copy() for b in range(B)]


# --------------------------------------------------------------------------
# torch-CPU fp32 reference for the floating-point networks
# --------------------------------------------------------------------------
def _t(x) -> torch.Tensor:
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def _slot_mlp(sd, prefix: str, center: torch.Tensor, cluster: torch.Tensor, training: bool = False) -> torch.Tensor:
    """Shared front of OffsetNetwork / SimplifiedPointNet (PRE:93-100, 131-138):
    per-slot 6 -> 256 point-wise conv + BatchNorm (eval: running statistics; training: batch statistics over all
    B*M*K slots, running statistics updated in ``sd``) + ReLU -> (B,M,K,256)."""
    rel = cluster - center[:, :, None, :]
    pad = (cluster == 0).all(dim=-1)                  # padded slot <=> xyz all zero (PRE:94)
    rel = torch.where(pad[..., None], torch.zeros_like(rel), rel)
    x = torch.cat([rel, cluster], dim=-1)             # (B,M,K,6)
    w = sd[prefix + ".mlp.0.weight"].reshape(-1, 6)
    h = F.linear(x, w, sd[prefix + ".mlp.0.bias"])
    h = F.batch_norm(h.permute(0, 3, 1, 2), sd[prefix + ".mlp.1.running_mean"],
                     sd[prefix + ".mlp.1.running_var"], sd[prefix + ".mlp.1.weight"],
                     sd[prefix + ".mlp.1.bias"], training=training, momentum=0.1, eps=BN_EPS)
    return F.relu(h).permute(0, 2, 3, 1)


def offset_net(sd, center, cluster, training: bool = False) -> torch.Tensor:
    """OffsetNetwork.forward, PRE:87-107 -> raw offsets (B,M,3) (before tanh)."""
    pre = "get_deformable_cluster.get_offsets"
    h = _slot_mlp(sd, pre, center, cluster, training).mean(dim=2)  # mean over K (PRE:102)
    return F.linear(h, sd[pre + ".channel_mapper.weight"].reshape(3, -1))


def point_encoder(sd, center, cluster, training: bool = False) -> torch.Tensor:
    """SimplifiedPointNet.forward, PRE:126-142 -> (B,M',256); max over all K slots."""
    return _slot_mlp(sd, "simple_encoder", center, cluster, training).max(dim=2)[0]


def img_proxy(sd, img_feat: torch.Tensor, heads: int) -> torch.Tensor:
    """get_img_proxy + AttentionPool2d, PRE:335-342, 154-177 -> (B,V,C).

    Written in the 'all queries' form of the reference (token 0 returned)."""
    B, V, Cin, H, W = img_feat.shape
    x = img_feat.reshape(B * V, Cin, H * W).permute(0, 2, 1)       # (BV, HW, Cin)
    C = sd["channel_mapper.weight"].shape[0]
    x = F.linear(x, sd["channel_mapper.weight"].reshape(C, Cin), sd["channel_mapper.bias"])
    x = torch.cat([x.mean(dim=1, keepdim=True), x], dim=1)         # prepend mean token
    x = x + sd["attn_pool2d.positional_embedding"][None]
    q = F.linear(x, sd["attn_pool2d.q_proj.weight"], sd["attn_pool2d.q_proj.bias"])
    k = F.linear(x, sd["attn_pool2d.k_proj.weight"], sd["attn_pool2d.k_proj.bias"])
    v = F.linear(x, sd["attn_pool2d.v_proj.weight"], sd["attn_pool2d.v_proj.bias"])
    T = x.shape[1]
    hd = C // heads
    q = q.reshape(-1, T, heads, hd).transpose(1, 2) * hd ** -0.5
    k = k.reshape(-1, T, heads, hd).transpose(1, 2)
    v = v.reshape(-1, T, heads, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v          # (BV,h,T,hd)
    a = a.transpose(1, 2).reshape(-1, T, C)
    out = F.linear(a, sd["attn_pool2d.c_proj.weight"], sd["attn_pool2d.c_proj.bias"])[:, 0]
    out = F.layer_norm(out, (C,), sd["norm_img.weight"], sd["norm_img.bias"], LN_EPS)
    return out.reshape(B, V, C)


def slot_bias(sd, pre: str, C: int) -> torch.Tensor:
    """Per-kept-slot learned bias of ProxyAttention, PRE:212-215 -> (M',C)."""
    s = int(C ** 0.5)
    if s * s != C:       # not runnable by the reference: next larger grid, first C entries
        s += 1
    b1 = F.interpolate(sd[pre + ".pb_bias"], size=(s, s), mode="bilinear")
    n = b1.shape[1]
    return (b1.reshape(n, -1) + (sd[pre + ".pc_bias"] + sd[pre + ".pr_bias"]).reshape(n, -1))[:, :C]


def proxy_block(sd, pre: str, x: torch.Tensor, proxy: torch.Tensor,
                mask: Optional[torch.Tensor], heads: int) -> Dict[str, torch.Tensor]:
    """One ProxyBlock in eval mode (dropout / drop-path are identity), PRE:206-257, 273-276."""
    B, n, C = x.shape
    L = proxy.shape[1]
    hd = C // heads
    scale = hd ** -0.5
    xn = F.layer_norm(x, (C,), sd[pre + ".norm1.weight"], sd[pre + ".norm1.bias"], LN_EPS)
    xb = xn + slot_bias(sd, pre + ".attn", C)[None]
    qkv = F.linear(xb, sd[pre + ".attn.qkv.weight"], sd.get(pre + ".attn.qkv.bias"))
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    pt = F.linear(proxy, sd[pre + ".attn.proxy_proj.weight"], sd[pre + ".attn.proxy_proj.bias"])

    def split(t, rows):
        return t.reshape(B, rows, heads, hd).permute(0, 2, 1, 3)
    q, k, v, pt = split(q, n), split(k, n), split(v, n), split(pt, L)
    pa = torch.softmax((pt * scale) @ k.transpose(-2, -1), dim=-1)      # proxy as query, no mask
    pv = pa @ v                                                          # (B,h,L,hd)
    qa = (q * scale) @ pt.transpose(-2, -1)                              # (B,h,n,L)
    if mask is not None:
        qa = qa.masked_fill(~mask[:, None, None, :], -1e9)               # PRE:243-247
    o = (torch.softmax(qa, dim=-1) @ pv).transpose(1, 2).reshape(B, n, C)
    attn_out = F.linear(o, sd[pre + ".attn.proj.weight"], sd[pre + ".attn.proj.bias"])
    x1 = x + attn_out
    xn2 = F.layer_norm(x1, (C,), sd[pre + ".norm2.weight"], sd[pre + ".norm2.bias"], LN_EPS)
    h = F.gelu(F.linear(xn2, sd[pre + ".mlp.fc1.weight"], sd[pre + ".mlp.fc1.bias"]))
    x2 = x1 + F.linear(h, sd[pre + ".mlp.fc2.weight"], sd[pre + ".mlp.fc2.bias"])
    return dict(qkv=qkv, pt=pt, pv=pv, attn=o, x1=x1, out=x2)


def _bn1d_eval(sd, pre: str, x: torch.Tensor, training: bool = False) -> torch.Tensor:
    """BatchNorm1d over the channel (last) dim of (B,m,c), PRE:446, 455 (eval: running statistics; training: batch
    statistics over the B*m rows)."""
    return F.batch_norm(x.transpose(-2, -1), sd[pre + ".running_mean"], sd[pre + ".running_var"],
                        sd[pre + ".weight"], sd[pre + ".bias"], training=training, momentum=0.1,
                        eps=BN_EPS).transpose(-2, -1)


# --------------------------------------------------------------------------
# full forward
# --------------------------------------------------------------------------
def forward(sd_np: Dict[str, np.ndarray], *, grid_size: int, dynamic_drop_radio: float,
            num_sub: int, num_heads: int, text_blocks: int, img_blocks: int,
            points: np.ndarray, text_feats: np.ndarray, text_mask: np.ndarray,
            img_feat: np.ndarray, order_override: Optional[np.ndarray] = None,
            centers_override: Optional[np.ndarray] = None, stop_after: Optional[str] = None,
            num_threads: Optional[int] = None) -> Dict[str, object]:
    """Eval-mode forward of the reference module; returns every intermediate.

    ``order_override`` replays a captured ``sorted_indices[:, :Mt]``;
    ``centers_override`` injects the clamped centres of ball query #2.
    """
    if num_threads is not None:
        torch.set_num_threads(num_threads)
    sd = {k: _t(v) for k, v in sd_np.items()}
    gs, K = grid_size, num_sub
    M = gs ** 3
    Mt = M - int(M * EMPTY_DROP)
    Mk = int(M * (1 - dynamic_drop_radio))
    out: Dict[str, object] = {}
    pts = _f32(points)
    B, N, _ = pts.shape

    with torch.no_grad():
        centers0, mn, mx = grid_centers(pts, gs)
        _, cluster1 = ball_query(centers0, pts, K)                         # PRE:56
        raw = offset_net(sd, _t(centers0), _t(cluster1))                   # PRE:58
        offsets = raw.tanh() * MARGIN                                      # PRE:59
        newc = _t(centers0) + offsets                                      # PRE:61
        clamped = torch.max(torch.min(newc, _t(mx)[:, None]), _t(mn)[:, None])   # PRE:62
        if centers_override is not None:
            clamped = _t(_f32(centers_override))
        idx2, cluster2 = ball_query(clamped.numpy(), pts, K)               # PRE:65
        out.update(centers0=centers0, mn=mn, mx=mx, cluster1=cluster1, offsets=offsets.numpy(),
                   centers=clamped.numpy(), idx2=idx2, cluster2=cluster2)
        if stop_after == "cluster":
            return out

        sel = select_clusters(idx2, clamped.numpy(), Mt, Mk, order_override)
        order, picks, keep = sel["order"], sel["picks"], sel["keep"]
        bidx = np.arange(B)[:, None]
        keep_src = order[bidx, keep]                                       # original cluster ids
        pick_src = order[bidx, picks]
        kcenter = clamped.numpy()[bidx, keep_src]
        kcluster = cluster2[bidx, keep_src]
        kidx = idx2[bidx, keep_src]
        drop_idx = idx2[bidx, pick_src].reshape(B, -1)                     # PRE:417-418
        out.update(pad_counts=sel["pad_counts"], order=order, picks=picks, keep=keep,
                   keep_src=keep_src, pick_src=pick_src, kcenter=kcenter, kcluster=kcluster,
                   kidx=kidx, drop_idx=drop_idx)
        if stop_after == "select":
            return out

        pp = point_encoder(sd, _t(kcenter), _t(kcluster))                  # PRE:437
        tb = proxy_block(sd, f"textformer.{text_blocks - 1}", pp, _t(_f32(text_feats)),
                         _t(np.asarray(text_mask, bool)), num_heads)       # PRE:441-442 (H8)
        C = pp.shape[-1]
        tg = F.layer_norm(tb["out"], (C,), sd[f"text_norm.{text_blocks - 1}.weight"],
                          sd[f"text_norm.{text_blocks - 1}.bias"], LN_EPS)
        translate = _bn1d_eval(sd, "text_trans_norm",
                               F.linear(tg, sd["text_trans.weight"], sd["text_trans.bias"]))
        ip = img_proxy(sd, _t(_f32(img_feat)), num_heads)                  # PRE:449
        ib = proxy_block(sd, f"imgformer.{img_blocks - 1}", pp, ip, None, num_heads)
        ig = F.layer_norm(ib["out"], (C,), sd[f"img_norm.{img_blocks - 1}.weight"],
                          sd[f"img_norm.{img_blocks - 1}.bias"], LN_EPS)
        transform = _bn1d_eval(sd, "img_trans_norm",
                               F.linear(ig, sd["img_trans.weight"], sd["img_trans.bias"]))
        T = transform.reshape(B, Mk, 3, 3)
        c = _t(kcenter)[:, :, None, :]
        newcl = (T @ (_t(kcluster) - c).transpose(-2, -1)).transpose(-2, -1) + c \
            + translate[:, :, None, :]                                     # PRE:459-462
        out.update(point_proxy=pp.numpy(), img_proxy=ip.numpy(), text_guide=tg.numpy(),
                   img_guide=ig.numpy(), translate=translate.numpy(),
                   transform=transform.numpy(), new_clusters=newcl.numpy(),
                   text_block=tb, img_block=ib)
        new_points = pt_replace(pts, kidx, newcl.numpy())                  # PRE:465
        out["new_points"] = new_points
        out["outputs"] = remove_points(new_points, drop_idx)               # PRE:467
    return out


# --------------------------------------------------------------------------
# train-mode forward + backward -- dropout / drop-path rates 0
# --------------------------------------------------------------------------
def loss_weights(b: int, n: int) -> np.ndarray:
    """Upstream gradient of output b used by the train-mode parity tests (same formula as tests/golden/gen_golden.py)."""
    i = np.arange(n, dtype=np.float64)[:, None]
    d = np.arange(3, dtype=np.float64)[None, :]
    return np.sin(0.37 * i + 1.3 * d + 0.7 * b).astype(np.float32)


def forward_train(sd_np: Dict[str, np.ndarray], *, grid_size: int, dynamic_drop_radio: float, num_sub: int,
                  num_heads: int, text_blocks: int, img_blocks: int, points: np.ndarray, text_feats: np.ndarray,
                  text_mask: np.ndarray, img_feat: np.ndarray, centers_override: Optional[np.ndarray] = None,
                  backward: bool = True, num_threads: Optional[int] = 1, float64: bool = False,
                  timing_only: bool = False) -> Dict[str, object]:
    """One training step of the reference module without the optimiser: train-mode forward (batch-statistics
    BatchNorm2d / BatchNorm1d with running-stat update, PRE:74, 114, 329-330; Dropout / DropPath at rate 0),
    loss = sum_b <out_b, loss_weights(b)>, backward through torch autograd.  The index steps are the C functions of
    the eval oracle (not differentiable, like pytorch3d's); pt_replace is torch's index_put_ (PRE:495), whose backward
    hands every valid slot the gradient of the point it targeted.  ``float64=True`` evaluates the float half in double
    (same formulas; the index half stays the fp32 C code): the gradients are then accurate far beyond what two fp32
    evaluations agree on, which is what the GPU gradients are held against.  Returns outputs, intermediates, ``grads``
    (name -> array, inputs as ``input.text_feats`` / ``input.img_feat``), ``none_grads`` and the updated buffers."""
    # index_put_ with duplicate targets (PRE:495) is last-writer-wins only single-threaded: this oracle
    # refuses any other setting rather than return an order-dependent scatter (oracle.forward's scatter is the C loop,
    # which is why ITS num_threads is free)
    # (``timing_only=True``: bench.py's cpu_baseline leg, which reads no values).
    if num_threads not in (None, 1) and not timing_only:
        raise ValueError("forward_train is deterministic with one torch thread only (index_put_, SURVEY H1)")
    torch.set_num_threads(num_threads if timing_only and num_threads else 1)
    ft = torch.float64 if float64 else torch.float32
    sd = {k: (_t(v).clone().to(ft) if np.asarray(v).dtype.kind == "f" else _t(v).clone()) for k, v in sd_np.items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    gs, K = grid_size, num_sub
    M = gs ** 3
    Mt = M - int(M * EMPTY_DROP)
    Mk = int(M * (1 - dynamic_drop_radio))
    pts = _f32(points)
    B, N, _ = pts.shape
    tf = _t(_f32(text_feats)).clone().to(ft).requires_grad_(True)
    im = _t(_f32(img_feat)).clone().to(ft).requires_grad_(True)
    out: Dict[str, object] = {}

    centers0, mn, mx = grid_centers(pts, gs)
    _, cluster1 = ball_query(centers0, pts, K)
    raw = offset_net(sd, _t(centers0).to(ft), _t(cluster1).to(ft), training=True)
    newc = _t(centers0).to(ft) + raw.tanh() * MARGIN
    clamped = torch.max(torch.min(newc, _t(mx).to(ft)[:, None]), _t(mn).to(ft)[:, None])          # PRE:62
    cq = _f32(clamped.detach().numpy()) if centers_override is None else _f32(centers_override)
    idx2, cluster2 = ball_query(cq, pts, K)
    sel = select_clusters(idx2, cq, Mt, Mk)
    order, picks, keep = sel["order"], sel["picks"], sel["keep"]
    bidx = np.arange(B)[:, None]
    keep_src, pick_src = order[bidx, keep], order[bidx, picks]
    kcenter = clamped[torch.from_numpy(bidx), torch.from_numpy(keep_src)]           # differentiable gather (PRE:414)
    kcluster = _t(cluster2[bidx, keep_src]).to(ft)
    kidx = idx2[bidx, keep_src]
    drop_idx = idx2[bidx, pick_src].reshape(B, -1)

    pp = point_encoder(sd, kcenter, kcluster, training=True)
    tb = proxy_block(sd, f"textformer.{text_blocks - 1}", pp, tf, _t(np.asarray(text_mask, bool)), num_heads)
    C = pp.shape[-1]
    tg = F.layer_norm(tb["out"], (C,), sd[f"text_norm.{text_blocks - 1}.weight"], sd[f"text_norm.{text_blocks - 1}.bias"], LN_EPS)
    translate = _bn1d_eval(sd, "text_trans_norm", F.linear(tg, sd["text_trans.weight"], sd["text_trans.bias"]), True)
    ip = img_proxy(sd, im, num_heads)
    ib = proxy_block(sd, f"imgformer.{img_blocks - 1}", pp, ip, None, num_heads)
    ig = F.layer_norm(ib["out"], (C,), sd[f"img_norm.{img_blocks - 1}.weight"], sd[f"img_norm.{img_blocks - 1}.bias"], LN_EPS)
    transform = _bn1d_eval(sd, "img_trans_norm", F.linear(ig, sd["img_trans.weight"], sd["img_trans.bias"]), True)
    T = transform.reshape(B, Mk, 3, 3)
    c = kcenter[:, :, None, :]
    newcl = (T @ (kcluster - c).transpose(-2, -1)).transpose(-2, -1) + c + translate[:, :, None, :]   # PRE:459-462
    # pt_replace (PRE:478-495) with torch's own index_put_ + remove_points_by_index (PRE:516-523)
    new_points = _t(pts).clone().to(ft)
    kidx_t = torch.from_numpy(kidx)
    valid = kidx_t != -1
    bi = torch.arange(B).reshape(B, 1, 1).expand(B, Mk, K)
    new_points[bi[valid], kidx_t[valid], :] = newcl[valid]
    outs = []
    for b in range(B):
        dropset = np.unique(drop_idx[b][drop_idx[b] >= 0])
        keepmask = np.ones(N, bool)
        keepmask[dropset] = False
        outs.append(new_points[b][torch.from_numpy(np.nonzero(keepmask)[0])])
    out.update(centers=clamped.detach().numpy(), idx2=idx2, order=order, picks=picks, keep=keep, kidx=kidx,
               drop_idx=drop_idx, point_proxy=pp.detach().numpy(), img_proxy=ip.detach().numpy(),
               translate=translate.detach().numpy(), transform=transform.detach().numpy(),
               outputs=[o.detach().numpy() for o in outs])
    if backward:
        loss = sum((o * torch.from_numpy(loss_weights(b, o.shape[0])).to(ft)).sum() for b, o in enumerate(outs))
        loss.backward()
        grads, none = {}, []
        for name, p_ in list(params.items()) + [("input.text_feats", tf), ("input.img_feat", im)]:
            if p_.grad is None:
                none.append(name)
            else:
                grads[name] = p_.grad.numpy()
        out.update(loss=float(loss.item()), grads=grads, none_grads=none)
    out["buffers"] = {k: v.detach().numpy() for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}
    return out


# --------------------------------------------------------------------------
# voxel quantisation right after the path -- PARITY UNPINNED against MinkowskiEngine
# --------------------------------------------------------------------------
def voxelize(outs, voxel_size: float = 0.01):
    """detectors/sparse_featfusion_grounder_preshape.py:388-397 restated: ``batch_sparse_collate`` floors
    ``p / voxel_size`` (fp32 division) to int32 and prepends the scene index; ``ME.SparseTensor`` keeps one row per
    distinct coordinate.  MinkowskiEngine is not vendored under /root/reference and its choice of the surviving
    duplicate / row order is unspecified ("random subsample"): pinned to the first point of every voxel in (scene,
    point) order.  Returns coords (Nv,4) int32, feats (Nv,3) f32, list of per-scene inverse maps."""
    coords, feats = [], []
    for b, p in enumerate(outs):
        p = _f32(p)
        v = np.floor(p / np.float32(voxel_size)).astype(np.int32)
        coords.append(np.concatenate([np.full((len(p), 1), b, np.int32), v], axis=1))
        feats.append(p)
    coords, feats = np.concatenate(coords), np.concatenate(feats)
    _, first, inv = np.unique(coords, axis=0, return_index=True, return_inverse=True)
    rows = np.sort(first)                                   # representatives in (scene, point) order
    rank = np.empty(len(first), np.int64)
    rank[np.argsort(first)] = np.arange(len(first))
    inverse = rank[inv.reshape(-1)].astype(np.int32)
    sizes = np.cumsum([0] + [len(p) for p in outs])
    return coords[rows], feats[rows], [inverse[sizes[b]:sizes[b + 1]] for b in range(len(outs))]


def level_coordinates(coords, n_scenes: int, stride: int):
    """Coordinates of a MinkowskiEngine level of output tensor stride ``stride`` over the voxel rows ``coords`` (Nv,4) int32
    (MinkResNet, backbones/mink_resnet.py:57-78: four stages at tensor strides 8 / 16 / 32 / 64; DET:398, 429-430 read
    ``x[level].decomposed_coordinates[idx]``): a strided ME layer maps a coordinate c to ``floor(c / stride) * stride`` and keeps
    one row per distinct result -- independent of the layer's weights.  ME is not vendored (parity unpinned, like ``voxelize``):
    row order pinned to first occurrence.  Returns a list of (n_b, 3) int32 arrays."""
    coords = np.asarray(coords, np.int32)
    out = []
    for b in range(n_scenes):
        c = coords[coords[:, 0] == b, 1:]
        lv = np.floor_divide(c, stride) * stride
        _, first = np.unique(lv, axis=0, return_index=True)
        out.append(lv[np.sort(first)].astype(np.int32))
    return out


# --------------------------------------------------------------------------
# image feature -> point sampling after the backbone
# --------------------------------------------------------------------------
def _fma32(a, b, c):
    """fp32 fused multiply-add, emulated: the product of two fp32 values is exact in double."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def reverse_3d_points(points, img_meta, coord_type="DEPTH"):
    """apply_3d_transformation(..., reverse=True) (point_fusion.py:20-107) step by step in fp32, as the reference applies
    it: the recorded flow back to front -- 'T': += -pcd_trans, 'S': *= 1 / pcd_scale_factor, 'R': @ inverse(pcd_rotation),
    'HF' / 'VF': the BEV flips of the coordinate type -- DEPTH negates x / y (depth_points.py:47-50), LIDAR y / x
    (lidar_points.py:47-50), CAMERA x / z (cam_points.py:47-50)."""
    p = torch.from_numpy(_f32(points)).clone()
    flow = list(img_meta.get("transformation_3d_flow", []))
    rot = torch.from_numpy(_f32(img_meta["pcd_rotation"])) if "pcd_rotation" in img_meta else torch.eye(3)
    scale = img_meta.get("pcd_scale_factor", 1.0)
    trans = torch.from_numpy(_f32(img_meta["pcd_trans"])) if "pcd_trans" in img_meta else torch.zeros(3)
    hf_axis, vf_axis = {"DEPTH": (0, 1), "LIDAR": (1, 0), "CAMERA": (0, 2)}[coord_type.upper()]
    for op in flow[::-1]:
        if op == "T":
            p += -trans
        elif op == "S":
            p *= 1.0 / scale
        elif op == "R":
            p = p @ rot.inverse()
        elif op == "HF":
            if img_meta.get("pcd_horizontal_flip", False):
                p[:, hf_axis] = -p[:, hf_axis]
        elif op == "VF":
            if img_meta.get("pcd_vertical_flip", False):
                p[:, vf_axis] = -p[:, vf_axis]
        else:
            raise AssertionError(op)
    return p.numpy()


def point_sample(points, feats, proj, *, scale=(1.0, 1.0), crop=(0.0, 0.0), flip=False, ori_w=0.0, pad_hw=(480.0, 640.0),
                 bilinear=False):
    """batch_point_sample (models/layers/fusion_layers/point_fusion.py:208-313) as the detector calls it
    (detectors/sparse_featfusion_grounder_preshape.py:428-444: nearest sampling, zeros padding, align_corners=True,
    valid_flag=True), for points already in the projectable frame (the reverse 3D augmentation of
    apply_3d_transformation is host-side metadata).  feats (V,C,H,W), proj (V,4,4) = intrinsic @ extrinsic.
    Per view: q = P [x y z 1]; (u, v) = q[:2] / max(q[2], 1e-3) (structures/bbox_3d/utils.py:324-327); image
    transform scale -> crop -> flip; normalise by the padded size; nearest pixel (round half to even); a view is valid
    when 0 < x < w_pad, 0 < y < h_pad and depth > 0; result = sum of ALL views' samples / max(#valid, 1), zero rows
    where no view is valid (point_fusion.py:299-311)."""
    p = _f32(points)
    f = _f32(feats)
    P = _f32(proj)
    V, C, H, W = f.shape
    N = p.shape[0]
    pad_h, pad_w = np.float32(pad_hw[0]), np.float32(pad_hw[1])
    acc = np.zeros((N, C), np.float32)
    nvalid = np.zeros(N, np.int64)
    one = np.float32(1.0)
    for v in range(V):
        q = []
        for r in range(3):
            t = p[:, 0] * P[v, r, 0]
            t = _fma32(p[:, 1], np.full(N, P[v, r, 1], np.float32), t)
            t = _fma32(p[:, 2], np.full(N, P[v, r, 2], np.float32), t)
            q.append(t + P[v, r, 3])
        z = np.maximum(q[2], np.float32(1e-3))
        cx = (q[0] / z) * np.float32(scale[0]) - np.float32(crop[0])
        cy = (q[1] / z) * np.float32(scale[1]) - np.float32(crop[1])
        if flip:
            cx = np.float32(ori_w) - cx
        nx = cx / pad_w * np.float32(2) - one
        ny = cy / pad_h * np.float32(2) - one
        fx = ((nx + one) / np.float32(2)) * np.float32(W - 1)
        fy = ((ny + one) / np.float32(2)) * np.float32(H - 1)
        if not bilinear:
            ix, iy = np.rint(fx), np.rint(fy)
            inb = (ix >= 0) & (ix <= W - 1) & (iy >= 0) & (iy <= H - 1)
            ixi, iyi = np.where(inb, ix, 0).astype(np.int64), np.where(inb, iy, 0).astype(np.int64)
            samp = f[v][:, iyi, ixi].T                         # (N,C)
            acc = acc + np.where(inb[:, None], samp, np.float32(0))
        else:
            # F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True): four neighbours weighted by the
            # opposite areas, neighbours outside the map contribute 0 (aligned=True, point_fusion.py:287-293)
            x0, y0 = np.floor(fx), np.floor(fy)
            wx1, wy1 = fx - x0, fy - y0
            wx0, wy0 = one - wx1, one - wy1
            s = np.zeros((N, C), np.float32)
            for dx, dy, w in ((0, 0, wx0 * wy0), (1, 0, wx1 * wy0), (0, 1, wx0 * wy1), (1, 1, wx1 * wy1)):
                xx, yy = x0 + dx, y0 + dy
                ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
                xi, yi = np.where(ok, xx, 0).astype(np.int64), np.where(ok, yy, 0).astype(np.int64)
                s = s + np.where(ok[:, None], f[v][:, yi, xi].T * w[:, None].astype(np.float32), np.float32(0))
            acc = acc + s
        nvalid += (cx < pad_w) & (cx > 0) & (cy < pad_h) & (cy > 0) & (q[2] > 0)
    out = acc / np.maximum(nvalid, 1)[:, None].astype(np.float32)
    out[nvalid == 0] = 0
    return out, nvalid


# ------------------------------------------------------------------------------ multi-view depth ingest
def points_img2cam(grid: np.ndarray, cam2img: np.ndarray) -> torch.Tensor:
    """structures/bbox_3d/utils.py:336-368 on fp32 tensors (the array converter turns the numpy inputs into tensors of
    their own dtype; ``pad_cam2img`` takes the dtype of the points: fp32)."""
    pts = torch.from_numpy(np.ascontiguousarray(grid, dtype=np.float32))
    cam = torch.from_numpy(np.ascontiguousarray(cam2img))
    xys = pts[:, :2]
    depths = pts[:, 2].view(-1, 1)
    unnormed = torch.cat([xys * depths, depths], dim=1)
    pad = torch.eye(4, dtype=xys.dtype)
    pad[:cam.shape[0], :cam.shape[1]] = cam
    inv_t = torch.inverse(pad).transpose(0, 1)
    homo = torch.cat([unnormed, xys.new_ones((unnormed.shape[0], 1))], dim=1)
    return torch.mm(homo, inv_t)[:, :3]


def ingest(depth_imgs: np.ndarray, depth_cam2img, extrinsic: np.ndarray, n_points: int, per_view: Optional[int] = None,
           rng=np.random, aug: Optional[dict] = None, num_threads: int = 1):
    """The reference pipeline between the decoded depth maps and the path's input cloud
    (configs/grounding/proxy-tiblock33-gs12-wbias-ddr0.6-clip.py:105-142), restated step by step:

      ConvertRGBDToPoints       datasets/transforms/points.py:57-69   meshgrid, float32 grid, nonzero filter, points_img2cam
      PointSample per view      points.py:335-336 (empty view: untouched), 395-411 (np.random.choice, replace iff too few)
      AggregateMultiViewPoints  multiview.py:224-239                  torch.linalg.solve(global2ego, [p;1]^T), concat
      PointSample of the scene  points.py:395-411
      GlobalRotScaleTrans       augmentation.py:327-352 on the points: p @ rot_mat_T, * scale, + trans (given parameters)

    depth_imgs (V,H,W) float32; depth_cam2img one matrix or (V,r,c); extrinsic (V,4,4) float32 global2ego.
    Returns dict(points (n_points,3) float32, view_counts, sel = the composed index per output point)."""
    torch.set_num_threads(num_threads)
    per_view = n_points // 10 if per_view is None else per_view
    V, H, W = depth_imgs.shape
    k = np.asarray(depth_cam2img)
    views, sels, counts = [], [], []
    off = 0
    for v in range(V):
        depth = np.ascontiguousarray(depth_imgs[v], dtype=np.float32)
        us, vs = np.meshgrid(np.arange(W), np.arange(H))
        grid = np.stack([us.astype(np.float32), vs.astype(np.float32), depth], axis=-1).reshape(-1, 3)
        nonzero = depth.reshape(-1).nonzero()[0]
        pts = points_img2cam(grid, k if k.ndim == 2 else k[v])[nonzero]
        counts.append(len(nonzero))
        if len(pts) > 0:                                         # PointSample returns an empty view untouched
            ch = rng.choice(np.arange(len(pts)), per_view, replace=len(pts) < per_view)
            pts = pts[ch]
            hom = torch.cat([pts, pts.new_ones(pts.shape[0], 1)], dim=1)
            g2e = torch.from_numpy(np.ascontiguousarray(extrinsic[v], dtype=np.float32))
            glob = torch.linalg.solve(g2e, hom.transpose(0, 1)).transpose(0, 1)
            views.append(glob[:, :3])
            sels.append(ch.astype(np.int64) + off)
        off += len(nonzero)
    cat = torch.cat(views)
    sel = np.concatenate(sels)
    ch2 = rng.choice(np.arange(len(cat)), n_points, replace=len(cat) < n_points)
    pts = cat[ch2].clone()
    if aug is not None:
        pts = pts @ torch.from_numpy(np.asarray(aug["rot_mat_T"], np.float32))
        pts *= float(aug["scale"])
        pts += torch.from_numpy(np.asarray(aug["trans"], np.float32))
    return dict(points=pts.numpy(), view_counts=np.asarray(counts, np.int64), sel=sel[ch2])
