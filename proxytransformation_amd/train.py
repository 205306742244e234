"""Train-mode forward + backward of ``ProxyTransformationNormReverse``.

The reference neck is trained (``runner.train()``, configs/grounding/proxy-tiblock33-gs12-wbias-ddr0.6-clip.py:200):
in train mode its BatchNorms use batch statistics (PRE:74, 114, 329-330), ProxyAttention / Mlp apply Dropout
(PRE:189-191, timm Mlp) and ProxyBlock applies DropPath (PRE:268), and gradients flow from the transformed points
back to every live parameter, to ``text_feats`` and to ``img_feat``.

How this file is built:

* the INDEX half of the path (grid centres, both ball queries, padding-count ordering, farthest point sampling, keep /
  drop lists, ownership tags, ordered compaction) runs through the same HIP stage entry points as eval mode and is not
  differentiable -- as in the reference, where pytorch3d returns integer indices (PRE:56, 65, 393);
* the FLOAT half is a chain of ``torch.autograd.Function`` nodes whose forward and backward bodies only launch
  hand-written HIP kernels through the C ABI (``ptx_op_*``, csrc/train_ops.hip).  torch owns the buffers and the graph
  (plumbing); no torch arithmetic operator is on the data path.  Fan-out is explicit (``_fork``) so that gradient
  sums are HIP launches too, and every parameter enters exactly one node whole;
* dead blocks (``textformer[:-1]``, ``imgformer[:-1]`` and their norms: PRE:441-443 feeds ``point_proxy`` to every
  block and keeps only the last output) are not executed: their parameters get ``grad = None`` exactly like
  in the reference (``find_unused_parameters=True``, CFG:246);
* gradients of ``pt_replace`` follow ``index_put_``'s backward (PRE:495): every valid slot whose target point survives
  the drop receives that point's output gradient, duplicates included.

Dropout masks come from a counter-based hash of (seed, call, site, element); torch's Philox stream is not reproduced
(not reproducible across devices anyway).  With all three rates at 0 the pass is deterministic and is
checked against gradients captured from the reference itself (tests/golden/g4_train.npz).
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch

from . import _abi

_F32 = torch.float32
# test switches: the generic one-launch-per-operator composition stays available (shapes outside the fused kernels' range use it)
# Which form of the step runs (module constants, not environment switches: tests/test_gpu_train.py patches them to drive the fallbacks).
_FUSED_ATTN = True       # the fused attention core of a block (ptx_train_attn_fwd / _bwd) instead of one launch per operator
_FUSED_BLOCK = True      # one ProxyBlock + trailing LayerNorm + head + BatchNorm1d as two C calls (ptx_train_block_fwd / _bwd)
_FUSED_IMG = True        # AttentionPool2d on its folded form (ptx_train_imgpool_fwd / _bwd)
_SIDE_STREAM = True      # the image branch on a side stream beside the index half
_BLOCKS_APART = True     # one-node step: the image block on the side stream too, beside the text block (forward and backward)
_ONE_NODE = True         # the float half as ONE autograd node (_TrainStep)
_C_STEP = True           # r06: the one node's two bodies behind ONE library call each (ptx_train_step_fwd / _bwd, csrc/train_step.hip)
_IMG_FIRST = True        # one-node step: the image branch enqueued in front of the clustering half (profiles/r04_train_ab.txt)
_IMG_POS = 1             # per-operator graph: where the image branch is enqueued -- 0 first, 1 after the selection, 2 before the
                         # text block, 3 after it


_TICKS = None          # scratch/train_hostprof3.py: list of (label, perf_counter) of the last forward


def _tick(label):
    if _TICKS is not None:
        import time
        _TICKS.append((label, time.perf_counter()))


def _side_stream(mod, dev):
    """The module's side stream for the image branch of the training step (one per device)."""
    cache = getattr(mod, "_train_side", None)
    if cache is None:
        cache = mod._train_side = {}
    st = cache.get(str(dev))
    if st is None:
        st = cache[str(dev)] = torch.cuda.Stream(device=dev)
    return st


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _st() -> int:
    # the raw handle of torch's current stream: ~500 launches per training step, and torch.cuda.current_stream() walks four
    # layers of Python (8 us) for each of them
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _ck(rc: int, what: str) -> None:
    if rc != 0:
        _abi.check(rc, what)


def _p(t: Optional[torch.Tensor], off: int = 0) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() if off == 0 else t.data_ptr() + off * t.element_size()      # ~300 calls per training step


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------- thin kernel wrappers
def gemm(A, B, C, M, N, K, a=(0, 1), b=(0, 1), c=(0, 1), batch=1, inner=1, a_bs=(0, 0), b_bs=(0, 0), c_bs=(0, 0),
         a_off=0, b_off=0, c_off=0, a_dtype=0, b_dtype=0, alpha=1.0, accumulate=False, ksplit=1, c_sk=0):
    """C[z][m][n] (+)= alpha sum_k A[z][m][k] B[z][k][n]; a / b / c = (row stride, col stride) in elements."""
    _ck(_abi.lib().ptx_op_gemm(_p(A, a_off), _p(B, b_off), _p(C, c_off), M, N, K, a[0], a[1], b[0], b[1], c[0], c[1],
                               batch, inner, a_bs[0], a_bs[1], b_bs[0], b_bs[1], c_bs[0], c_bs[1], a_dtype, b_dtype, alpha,
                               1 if accumulate else 0, ksplit, c_sk, _st()), "ptx_op_gemm")


def mm(a2, b2, ta=False, tb=False, out=None, alpha=1.0, accumulate=False):
    """Dense 2-D product of contiguous matrices: op(a2) @ op(b2)."""
    ar, ac = a2.shape
    br, bc = b2.shape
    M, K = (ac, ar) if ta else (ar, ac)
    K2, N = (bc, br) if tb else (br, bc)
    assert K == K2, (a2.shape, b2.shape, ta, tb)
    if out is None:
        out = torch.empty((M, N), dtype=_F32, device=a2.device)
    kw = dict(a=(1, ac) if ta else (ac, 1), b=(1, bc) if tb else (bc, 1), c=(N, 1), alpha=alpha)
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    if K >= 1024 and tiles < 512 and not accumulate:
        # a long contraction into a small result (weight gradients): slices of K across the chip, summed in order.  A
        # work-group's walk over K is a chain of dependent round trips, so the slices are short (>= 256 k = 8 steps) and many
        ks = int(min(256, max(2, 2048 // tiles), max(2, K // 256)))
        part = torch.empty((ks, M * N), dtype=_F32, device=a2.device)
        gemm(a2, b2, part, M, N, K, ksplit=ks, c_sk=M * N, **kw)
        colsum(part, out=out.view(-1))
        return out
    gemm(a2, b2, out, M, N, K, accumulate=accumulate, **kw)
    return out


def colsum(x2, y2=None, mode=0, scale=1.0, out=None, accumulate=False):
    R, N = x2.shape
    if out is None:
        out = torch.empty((N,), dtype=_F32, device=x2.device)
    # enough (column tile, row split) blocks to fill the chip; every split walks >= ~64 rows per slice
    nsplit = int(max(1, min(512, R // 32, 4096 // ((N + 255) // 256))))
    scratch = torch.empty((nsplit, N), dtype=torch.float64, device=x2.device)
    _ck(_abi.lib().ptx_op_colsum(_p(x2), _p(y2), R, N, mode, scale, 1 if accumulate else 0, _p(out), _p(scratch), nsplit,
                                 _st()), "ptx_op_colsum")
    return out


def eltwise(op, a, b=None, s=0.0, ncol=1, out=None):
    if out is None:
        out = torch.empty_like(a)
    _ck(_abi.lib().ptx_op_eltwise(op, _p(a), _p(b), s, a.numel(), ncol, _p(out), _st()), "ptx_op_eltwise")
    return out


def add_(a, b):
    return eltwise(0, a, b)


def dropout_k(x, p, seed, group=1):
    y = torch.empty_like(x)
    _ck(_abi.lib().ptx_op_dropout(_p(x), x.numel(), group, p, seed & 0xFFFFFFFFFFFFFFFF, _p(y), _st()), "ptx_op_dropout")
    return y


# --------------------------------------------------------------------------- autograd nodes
class _StreamHop(torch.autograd.Function):
    """Identity on the forward stream; its backward (same stream) tells the allocator that the gradient it passes on is about to be
    READ ON ANOTHER STREAM.  The image branch of the per-operator graph runs on a side stream: the gradient of its output is allocated
    by the image block's backward on the main stream, consumed by the side stream's nodes and dropped as soon as the first of them
    returns -- without the record the block goes back to the main stream's pool while the side stream still reads it (r05: the
    gradients of the image pool's parameters came out different from run to run once a second loss term changed the order in which
    the engine ran its nodes).  The one-node step records its hand-over itself."""

    @staticmethod
    def forward(ctx, x, stream):
        ctx.stream = stream
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        g.record_stream(ctx.stream)
        return g, None


class _Fork(torch.autograd.Function):
    """n handles of one tensor; the backward sums the n gradients with HIP launches (no torch add)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        live = [_c(g) for g in grads if g is not None]
        if not live:
            return None, None
        acc = live[0]
        for g in live[1:]:
            acc = add_(acc, g)
        return acc, None


def fork(x, n):
    return _Fork.apply(x, n)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return add_(_c(a), _c(b))

    @staticmethod
    def backward(ctx, g):
        return g, g


def _tuned_ok(*tensors_and_k):
    """The eval path's tuned NT GEMM (ptx_linear) takes dense fp32 operands with n_in % 4 == 0, 16-byte aligned."""
    *ts, k = tensors_and_k
    return k % 4 == 0 and k >= 4 and all(t.data_ptr() % 16 == 0 for t in ts)


def linear_nt(x, w2, b=None):
    """x (R,K) @ w2 (N,K)^T (+ b): the tuned MFMA kernel where its layout rules hold, the strided one otherwise."""
    R, K = x.shape
    N = w2.shape[0]
    if _tuned_ok(x, w2, K):
        y = torch.empty((R, N), dtype=_F32, device=x.device)
        _ck(_abi.lib().ptx_linear(_p(x), _p(w2), _p(b), None, _p(y), R, N, K, 0, _st()), "ptx_linear")
        return y
    y = mm(x, w2, tb=True)
    if b is not None:
        eltwise(6, y, b, ncol=N, out=y)
    return y


def transpose2d(x2):
    r, c = x2.shape
    out = torch.empty((c, r), dtype=_F32, device=x2.device)
    _ck(_abi.lib().ptx_op_transpose(_p(x2), r, c, _p(out), _st()), "ptx_op_transpose")
    return out


class _Linear(torch.autograd.Function):
    """y = x w^T + b (nn.Linear / 1x1 convolutions); w may be any contiguous tensor whose leading dim is n_out."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _c(x)
        w2 = w.reshape(w.shape[0], -1)
        y = linear_nt(x, w2, b)
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        w2 = w.reshape(w.shape[0], -1)
        dx = None
        if ctx.needs_input_grad[0]:     # dy (R,N) @ w2 (N,K) = dy @ (w2^T)^T: NT again with the transposed weight
            dx = linear_nt(dy, transpose2d(w2)) if _tuned_ok(dy, dy.shape[1]) and w2.shape[1] % 4 == 0 else mm(dy, w2)
        dw = mm(dy, x, ta=True).view_as(w) if ctx.needs_input_grad[1] else None
        db = colsum(dy) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        return dx, dw, db


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = _c(x)
        R, C = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((R, 2), dtype=_F32, device=x.device)
        _ck(_abi.lib().ptx_op_layernorm_fwd(_p(x), _p(w), _p(b), None, 1, R, C, eps, _p(y), _p(stats), _st()), "layernorm_fwd")
        ctx.save_for_backward(x, w, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, stats = ctx.saved_tensors
        dy = _c(dy)
        R, C = x.shape
        dx = torch.empty_like(x)
        xhat = torch.empty_like(x)
        _ck(_abi.lib().ptx_op_layernorm_bwd(_p(x), _p(w), _p(dy), _p(stats), R, C, _p(dx), _p(xhat), _st()), "layernorm_bwd")
        return dx, colsum(dy, xhat, mode=1), colsum(dy), None


class _SlotBiasAdd(torch.autograd.Function):
    """x + bias[j] with the learned per-slot table of ProxyAttention (PRE:212-217); j = row % Mk."""

    @staticmethod
    def forward(ctx, x, pb, pc, pr, Mk, s):
        x = _c(x)
        C = x.shape[1]
        table = torch.empty((Mk, C), dtype=_F32, device=x.device)
        _ck(_abi.lib().ptx_op_slotbias_fwd(_p(pb), _p(pc), _p(pr), Mk, s, C, _p(table), _st()), "slotbias_fwd")
        ctx.dims = (Mk, s, C)
        ctx.shapes = (pb.shape, pc.shape, pr.shape)
        return eltwise(6, x, table, ncol=Mk * C)

    @staticmethod
    def backward(ctx, dy):
        Mk, s, C = ctx.dims
        dy = _c(dy)
        dtab = colsum(dy.view(-1, Mk * C))
        dpb = torch.empty(ctx.shapes[0], dtype=_F32, device=dy.device)
        dpc = torch.empty(ctx.shapes[1], dtype=_F32, device=dy.device)
        dpr = torch.empty(ctx.shapes[2], dtype=_F32, device=dy.device)
        _ck(_abi.lib().ptx_op_slotbias_bwd(_p(dtab), Mk, s, C, _p(dpb), _p(dpc), _p(dpr), _st()), "slotbias_bwd")
        return dy, dpb, dpc, dpr, None, None


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return eltwise(2, x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return eltwise(3, x, _c(dy))


class _Dropout(torch.autograd.Function):
    """Dropout (group 1) / DropPath (group = elements per sample); identity node when p == 0."""

    @staticmethod
    def forward(ctx, x, p, seed, group):
        ctx.cfg = (p, seed, group)
        return dropout_k(_c(x), p, seed, group)

    @staticmethod
    def backward(ctx, dy):
        p, seed, group = ctx.cfg
        return dropout_k(_c(dy), p, seed, group), None, None, None


def dropout(x, p, seed, group=1):
    return x if p <= 0.0 else _Dropout.apply(x, float(p), int(seed), int(group))


class _BatchNormRows(torch.autograd.Function):
    """BatchNorm over the rows of (R,C) with batch statistics (+ optional fused ReLU); updates the running stats."""

    @staticmethod
    def forward(ctx, x, w, b, run_mean, run_var, eps, momentum, relu):
        x = _c(x)
        R, C = x.shape
        lib = _abi.lib()
        s1 = colsum(x, scale=1.0 / R)                           # mean, then the sum of squares around it
        s2 = colsum(x, s1, mode=3)
        mr = torch.empty((2, C), dtype=_F32, device=x.device)
        _ck(lib.ptx_op_bn_stats(_p(s1), _p(s2), C, R, eps, momentum, _p(mr), _p(run_mean), _p(run_var), _st()), "bn_stats")
        y = torch.empty_like(x)
        _ck(lib.ptx_op_bn_apply(_p(x), _p(mr), _p(w), _p(b), R, C, 1 if relu else 0, _p(y), _st()), "bn_apply")
        ctx.save_for_backward(x, y if relu else x, mr, w)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mr, w = ctx.saved_tensors
        dy = _c(dy)
        R, C = x.shape
        lib = _abi.lib()
        g, gx = torch.empty_like(x), torch.empty_like(x)
        _ck(lib.ptx_op_bn_bwd_prep(_p(x), _p(y), _p(dy), _p(mr), R, C, 1 if ctx.relu else 0, _p(g), _p(gx), _st()), "bn_bwd_prep")
        dbeta, dgamma = colsum(g), colsum(gx)
        dx = torch.empty_like(x)
        _ck(lib.ptx_op_bn_bwd_dx(_p(x), _p(g), _p(mr), _p(w), _p(dbeta), _p(dgamma), R, C, _p(dx), _st()), "bn_bwd_dx")
        return dx, dgamma, dbeta, None, None, None, None, None


class _SlotNet(torch.autograd.Function):
    """Shared body of OffsetNetwork / SimplifiedPointNet (PRE:87-102, 126-140): slot inputs -> Conv2d(6,W,1) ->
    BatchNorm2d (batch statistics over all B*M*K slots) -> ReLU -> mean / max over the K slots."""

    @staticmethod
    def forward(ctx, center, cluster, conv_w, conv_b, bn_w, bn_b, run_mean, run_var, eps, momentum, maxpool):
        # fused (csrc/slotnet_train.hip): the (B*M*K, W) activations are recomputed from the six slot inputs in every pass
        # instead of being written and re-read (r02: GEMM + bias + two column sums + normalise + pool over 318 MB)
        lib = _abi.lib()
        center, cluster = _c(center), _c(cluster)
        n, K = center.shape[0], cluster.shape[-2]
        W = conv_w.shape[0]
        dev = center.device
        out = torch.empty((n, W), dtype=_F32, device=dev)
        arg = torch.empty((n, W), dtype=torch.int32, device=dev) if maxpool else None
        mr = torch.empty((2, W), dtype=_F32, device=dev)
        tmp = torch.empty((2, W), dtype=_F32, device=dev)
        scratch = torch.empty((lib.ptx_op_slotnet_scratch_bytes(W),), dtype=torch.uint8, device=dev)
        _ck(lib.ptx_op_slotnet_fwd(_p(center), _p(cluster), n, K, W, _p(conv_w), _p(conv_b), _p(bn_w), _p(bn_b), eps, momentum,
                                   _p(run_mean), _p(run_var), 1 if maxpool else 0, _p(out), _p(arg), _p(mr), _p(tmp),
                                   _p(scratch), scratch.numel(), _st()), "slotnet_fwd")
        ctx.save_for_backward(center, cluster, conv_w, conv_b, bn_w, bn_b, mr, arg if maxpool else mr)
        ctx.cfg = (n, K, W, maxpool)
        return out

    @staticmethod
    def backward(ctx, dout):
        center, cluster, conv_w, conv_b, bn_w, bn_b, mr, arg = ctx.saved_tensors
        n, K, W, maxpool = ctx.cfg
        lib = _abi.lib()
        dout = _c(dout)
        dev = dout.device
        dconv_w = torch.empty_like(conv_w)
        dconv_b = torch.empty((W,), dtype=_F32, device=dev)
        dbg = torch.empty((2, W), dtype=_F32, device=dev)
        dcenter = torch.empty((n, 3), dtype=_F32, device=dev) if ctx.needs_input_grad[0] else None
        scratch = torch.empty((lib.ptx_op_slotnet_scratch_bytes(W),), dtype=torch.uint8, device=dev)
        _ck(lib.ptx_op_slotnet_bwd(_p(center), _p(cluster), n, K, W, _p(conv_w), _p(conv_b), _p(bn_w), _p(bn_b), _p(mr),
                                   1 if maxpool else 0, _p(arg) if maxpool else None, _p(dout), _p(dconv_w), _p(dconv_b),
                                   _p(dbg), _p(dcenter), _p(scratch), scratch.numel(), _st()), "slotnet_bwd")
        return dcenter, None, dconv_w, dconv_b, dbg[1], dbg[0], None, None, None, None, None


class _OffsetHead(torch.autograd.Function):
    """mean-pooled features -> Conv1d(256,3,1,bias=False) -> tanh * margin -> + grid centre -> clamp (PRE:59-62, 103)."""

    @staticmethod
    def forward(ctx, pooled, map_w, c0, minmax, M, margin):
        pooled = _c(pooled)
        n = pooled.shape[0]
        raw = mm(pooled, map_w.reshape(3, -1), tb=True)
        cout = torch.empty((n, 3), dtype=_F32, device=pooled.device)
        dcoef = torch.empty_like(cout)
        _ck(_abi.lib().ptx_op_offset_apply(_p(c0), _p(raw), _p(minmax), n, M, margin, _p(cout), _p(dcoef), _st()), "offset_apply")
        ctx.save_for_backward(pooled, map_w, dcoef)
        return cout

    @staticmethod
    def backward(ctx, dc):
        pooled, map_w, dcoef = ctx.saved_tensors
        w2 = map_w.reshape(3, -1)
        n, W = pooled.shape
        if W % 64 == 0 and W <= 512:        # one pass: dpooled rows and the chunk partials of dmap_w, then their fixed-order sum
            lib = _abi.lib()
            dpooled = torch.empty_like(pooled)
            dw = torch.empty((3, W), dtype=_F32, device=pooled.device)
            nt = lib.ptx_op_head_bwd_tmp_floats(n, W, 3)
            tmp = torch.empty((nt,), dtype=_F32, device=pooled.device)
            _ck(lib.ptx_op_head_bwd(_p(_c(dc)), _p(dcoef), _p(pooled), _p(_c(w2)), n, W, 3, _p(dpooled), _p(dw), _p(tmp), nt, _st()),
                "ptx_op_head_bwd")
            return dpooled, dw.view_as(map_w), None, None, None, None
        draw = eltwise(8, _c(dc), dcoef)
        return mm(draw, w2), mm(draw, pooled, ta=True).view_as(map_w), None, None, None, None


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, src):
        x = _c(x)
        rows, C = src.shape[0], x.shape[1]
        y = torch.empty((rows, C), dtype=_F32, device=x.device)
        _ck(_abi.lib().ptx_op_rows_gather(_p(x), _p(src), rows, C, _p(y), _st()), "rows_gather")
        ctx.save_for_backward(src)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        (src,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, dtype=_F32, device=dy.device)          # memset, rows that were not kept get 0
        _ck(_abi.lib().ptx_op_rows_scatter(_p(_c(dy)), _p(src), src.shape[0], ctx.shape[1], _p(dx), _st()), "rows_scatter")
        return dx, None


class _ProxyAttnCore(torch.autograd.Function):
    """Both contractions of ProxyAttention (PRE:230-252) on head-split views of qkv (B*n,3C) and the projected proxies
    (B*L,C): proxy-as-query softmax over the n tokens (unmasked), proxy-as-key softmax over the L proxies (padded text
    tokens filled with -1e9), attention dropout on both maps."""

    @staticmethod
    def forward(ctx, qkv, pt, mask, B, n, L, heads, p_drop, seed):
        qkv, pt = _c(qkv), _c(pt)
        C = pt.shape[1]
        hd = C // heads
        scale = float(hd) ** -0.5
        dev = qkv.device
        lib = _abi.lib()
        Z = B * heads
        ctx.fused = False
        if _FUSED_ATTN and lib.ptx_train_attn_tmp_floats(B, n, L, heads, C) > 0:
            # five kernels (csrc/train_fused.hip) instead of 22 launches of small batched products, soft-maxes and dropouts
            P1 = torch.empty((B, heads, L, n), dtype=_F32, device=dev)
            PV = torch.empty((B, heads, L, hd), dtype=_F32, device=dev)
            P2 = torch.empty((B, heads, n, L), dtype=_F32, device=dev)
            O = torch.empty((B * n, C), dtype=_F32, device=dev)
            _ck(lib.ptx_train_attn_fwd(_p(qkv), _p(pt), _p(mask), B, n, L, heads, C, p_drop, seed & 0xFFFFFFFFFFFFFFFF, _p(P1),
                                       _p(PV), _p(P2), _p(O), _st()), "ptx_train_attn_fwd")
            ctx.save_for_backward(qkv, pt, P1, PV, P2, mask if mask is not None else P1)
            ctx.cfg = (B, n, L, heads, C, hd, scale, p_drop, seed, mask is not None)
            ctx.fused = True
            return O
        q = dict(a=(3 * C, 1), a_bs=(n * 3 * C, hd))            # views of qkv as the A operand
        # S1[b,h,l,i] = scale Pt[b,l,h,:] . K[b,i,h,:]
        S1 = torch.empty((B, heads, L, n), dtype=_F32, device=dev)
        gemm(pt, qkv, S1, L, n, hd, a=(C, 1), b=(1, 3 * C), c=(n, 1), batch=Z, inner=heads, a_bs=(L * C, hd),
             b_bs=(n * 3 * C, hd), c_bs=(heads * L * n, L * n), b_off=C, alpha=scale)
        P1 = torch.empty_like(S1)
        _ck(lib.ptx_op_softmax_fwd(_p(S1), None, Z * L, n, 1, _p(P1), _st()), "softmax_fwd")
        D1 = P1 if p_drop <= 0 else dropout_k(P1, p_drop, seed)
        PV = torch.empty((B, heads, L, hd), dtype=_F32, device=dev)
        gemm(D1, qkv, PV, L, hd, n, a=(n, 1), b=(3 * C, 1), c=(hd, 1), batch=Z, inner=heads, a_bs=(heads * L * n, L * n),
             b_bs=(n * 3 * C, hd), c_bs=(heads * L * hd, L * hd), b_off=2 * C)
        # S2[b,h,i,l] = scale Q[b,i,h,:] . Pt[b,l,h,:]
        S2 = S1.view(B, heads, n, L)                              # same size, reused
        gemm(qkv, pt, S2, n, L, hd, a=q["a"], b=(1, C), c=(L, 1), batch=Z, inner=heads, a_bs=q["a_bs"],
             b_bs=(L * C, hd), c_bs=(heads * n * L, n * L), alpha=scale)
        P2 = torch.empty((B, heads, n, L), dtype=_F32, device=dev)
        _ck(lib.ptx_op_softmax_fwd(_p(S2), _p(mask), Z * n, L, heads * n, _p(P2), _st()), "softmax_fwd")
        D2 = P2 if p_drop <= 0 else dropout_k(P2, p_drop, seed + 1)
        O = torch.empty((B * n, C), dtype=_F32, device=dev)
        gemm(D2, PV, O, n, hd, L, a=(L, 1), b=(hd, 1), c=(C, 1), batch=Z, inner=heads, a_bs=(heads * n * L, n * L),
             b_bs=(heads * L * hd, L * hd), c_bs=(n * C, hd))
        ctx.save_for_backward(qkv, pt, P1, D1, PV, P2, D2, mask if mask is not None else P1)
        ctx.cfg = (B, n, L, heads, C, hd, scale, p_drop, seed, mask is not None)
        return O

    @staticmethod
    def backward(ctx, dO):
        B, n, L, heads, C, hd, scale, p_drop, seed, has_mask = ctx.cfg
        dO = _c(dO)
        dev = dO.device
        lib = _abi.lib()
        if ctx.fused:
            qkv, pt, P1, PV, P2, mask = ctx.saved_tensors
            dqkv, dpt = torch.empty_like(qkv), torch.empty_like(pt)
            nt = lib.ptx_train_attn_tmp_floats(B, n, L, heads, C)
            tmp = torch.empty((nt,), dtype=_F32, device=dev)
            _ck(lib.ptx_train_attn_bwd(_p(qkv), _p(pt), _p(mask) if has_mask else None, B, n, L, heads, C, p_drop,
                                       seed & 0xFFFFFFFFFFFFFFFF, _p(P1), _p(PV), _p(P2), _p(dO), _p(dqkv), _p(dpt), _p(tmp), nt,
                                       _st()), "ptx_train_attn_bwd")
            return dqkv, dpt, None, None, None, None, None, None, None
        qkv, pt, P1, D1, PV, P2, D2, mask = ctx.saved_tensors
        Z = B * heads
        dqkv = torch.empty_like(qkv)
        dpt = torch.empty_like(pt)
        # dD2[b,h,i,l] = dO[b,i,h,:] . PV[b,h,l,:] ;  dPV[b,h,l,:] = sum_i D2[b,h,i,l] dO[b,i,h,:]
        dD2 = torch.empty_like(P2)
        gemm(dO, PV, dD2, n, L, hd, a=(C, 1), b=(1, hd), c=(L, 1), batch=Z, inner=heads, a_bs=(n * C, hd),
             b_bs=(heads * L * hd, L * hd), c_bs=(heads * n * L, n * L))
        dPV = torch.empty_like(PV)
        gemm(D2, dO, dPV, L, hd, n, a=(1, L), b=(C, 1), c=(hd, 1), batch=Z, inner=heads, a_bs=(heads * n * L, n * L),
             b_bs=(n * C, hd), c_bs=(heads * L * hd, L * hd))
        dP2 = dD2 if p_drop <= 0 else dropout_k(dD2, p_drop, seed + 1)
        dS2 = torch.empty_like(P2)
        _ck(lib.ptx_op_softmax_bwd(_p(P2), _p(dP2), _p(mask) if has_mask else None, Z * n, L, heads * n, _p(dS2), _st()), "softmax_bwd")
        # dQ[b,i,h,:] = scale sum_l dS2[b,h,i,l] Pt[b,l,h,:]  -> columns 0..C of dqkv
        gemm(dS2, pt, dqkv, n, hd, L, a=(L, 1), b=(C, 1), c=(3 * C, 1), batch=Z, inner=heads, a_bs=(heads * n * L, n * L),
             b_bs=(L * C, hd), c_bs=(n * 3 * C, hd), alpha=scale)
        # dPt[b,l,h,:] = scale sum_i dS2[b,h,i,l] Q[b,i,h,:]
        gemm(dS2, qkv, dpt, L, hd, n, a=(1, L), b=(3 * C, 1), c=(C, 1), batch=Z, inner=heads, a_bs=(heads * n * L, n * L),
             b_bs=(n * 3 * C, hd), c_bs=(L * C, hd), alpha=scale)
        # dD1[b,h,l,i] = dPV[b,h,l,:] . V[b,i,h,:] ;  dV[b,i,h,:] = sum_l D1[b,h,l,i] dPV[b,h,l,:]  -> columns 2C..3C
        dD1 = torch.empty_like(P1)
        gemm(dPV, qkv, dD1, L, n, hd, a=(hd, 1), b=(1, 3 * C), c=(n, 1), batch=Z, inner=heads, a_bs=(heads * L * hd, L * hd),
             b_bs=(n * 3 * C, hd), c_bs=(heads * L * n, L * n), b_off=2 * C)
        gemm(D1, dPV, dqkv, n, hd, L, a=(1, n), b=(hd, 1), c=(3 * C, 1), batch=Z, inner=heads, a_bs=(heads * L * n, L * n),
             b_bs=(heads * L * hd, L * hd), c_bs=(n * 3 * C, hd), c_off=2 * C)
        dP1 = dD1 if p_drop <= 0 else dropout_k(dD1, p_drop, seed)
        dS1 = torch.empty_like(P1)
        _ck(lib.ptx_op_softmax_bwd(_p(P1), _p(dP1), None, Z * L, n, 1, _p(dS1), _st()), "softmax_bwd")
        # dPt += scale sum_i dS1[b,h,l,i] K[b,i,h,:] ;  dK[b,i,h,:] = scale sum_l dS1[b,h,l,i] Pt[b,l,h,:]  -> columns C..2C
        gemm(dS1, qkv, dpt, L, hd, n, a=(n, 1), b=(3 * C, 1), c=(C, 1), batch=Z, inner=heads, a_bs=(heads * L * n, L * n),
             b_bs=(n * 3 * C, hd), c_bs=(L * C, hd), b_off=C, alpha=scale, accumulate=True)
        gemm(dS1, pt, dqkv, n, hd, L, a=(1, n), b=(C, 1), c=(3 * C, 1), batch=Z, inner=heads, a_bs=(heads * L * n, L * n),
             b_bs=(L * C, hd), c_bs=(n * 3 * C, hd), c_off=C, alpha=scale)
        return dqkv, dpt, None, None, None, None, None, None, None


class _ImgTokens(torch.autograd.Function):
    """Conv2d(in_dim,C,1) over every pixel + mean token + positional embedding (PRE:338, 155-157) -> (nimg, hw+1, C)."""

    @staticmethod
    def forward(ctx, img, wc, bc, pos):
        # img (nimg, Cin, hw) in its storage type (fp32 / bf16 / fp16)
        nimg, Cin, hw = img.shape
        C = wc.shape[0]
        a_dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[img.dtype]
        tok = torch.zeros((nimg, hw + 1, C), dtype=_F32, device=img.device)
        gemm(img, wc, tok, hw, C, Cin, a=(1, hw), b=(1, Cin), c=(C, 1), batch=nimg, inner=1, a_bs=(Cin * hw, 0),
             b_bs=(0, 0), c_bs=((hw + 1) * C, 0), c_off=C, a_dtype=a_dt)
        eltwise(6, tok, bc, ncol=C, out=tok)                     # row 0 is overwritten below
        _ck(_abi.lib().ptx_op_tokens_finish(_p(tok), _p(pos), nimg, hw, C, _st()), "tokens_finish")
        ctx.save_for_backward(img, wc)
        ctx.a_dt = a_dt
        return tok

    @staticmethod
    def backward(ctx, dtok):
        img, wc = ctx.saved_tensors
        nimg, Cin, hw = img.shape
        C = wc.shape[0]
        dtok = _c(dtok)
        dpos = colsum(dtok.view(nimg, (hw + 1) * C)).view(hw + 1, C)
        d2 = eltwise(1, dtok, s=1.0)                             # private copy: the mean-token gradient is folded in place
        _ck(_abi.lib().ptx_op_tokens_finish_bwd(_p(d2), nimg, hw, C, _st()), "tokens_finish_bwd")
        dbc = colsum(d2.view(nimg * (hw + 1), C))                # token-0 rows are zero now
        # dWc[c][cin] = sum_img sum_p d2[img][1+p][c] img[img][cin][p]: per-image products, then a fixed-order sum over images
        part = torch.empty((nimg, C * Cin), dtype=_F32, device=dtok.device)
        gemm(d2, img, part, C, Cin, hw, a=(1, C), b=(1, hw), c=(Cin, 1), batch=nimg, inner=1, a_bs=((hw + 1) * C, 0),
             b_bs=(Cin * hw, 0), c_bs=(C * Cin, 0), a_off=C, b_dtype=ctx.a_dt)
        dwc = colsum(part).view_as(wc)
        dimg = None
        if ctx.needs_input_grad[0]:
            dimg32 = torch.empty((nimg, Cin, hw), dtype=_F32, device=dtok.device)
            gemm(wc, d2, dimg32, Cin, hw, C, a=(1, Cin), b=(1, C), c=(hw, 1), batch=nimg, inner=1, a_bs=(0, 0),
                 b_bs=((hw + 1) * C, 0), c_bs=(Cin * hw, 0), b_off=C)
            dimg = dimg32 if img.dtype == torch.float32 else dimg32.to(img.dtype)
        return dimg, dwc, dbc, dpos


class _AttnPoolCore(torch.autograd.Function):
    """AttentionPool2d's attention for the one query row that is returned (token 0, PRE:158-177): separate q / k / v
    projections, softmax over the hw+1 tokens, no dropout (dropout_p = 0 at PRE:169)."""

    @staticmethod
    def forward(ctx, tok, wq, bq, wk, bk, wv, bv, heads):
        tok = _c(tok)
        nimg, T, C = tok.shape
        hd = C // heads
        scale = float(hd) ** -0.5
        dev = tok.device
        tok2 = tok.view(nimg * T, C)
        Kt = linear_nt(tok2, wk, bk)
        Vt = linear_nt(tok2, wv, bv)
        q = torch.empty((nimg, C), dtype=_F32, device=dev)
        gemm(tok, wq, q, nimg, C, C, a=(T * C, 1), b=(1, C), c=(C, 1))
        eltwise(6, q, bq, ncol=C, out=q)
        Z = nimg * heads
        S = torch.empty((nimg, heads, T), dtype=_F32, device=dev)
        gemm(q, Kt, S, 1, T, hd, a=(0, 1), b=(1, C), c=(T, 1), batch=Z, inner=heads, a_bs=(C, hd), b_bs=(T * C, hd),
             c_bs=(heads * T, T), alpha=scale)
        P = torch.empty_like(S)
        _ck(_abi.lib().ptx_op_softmax_fwd(_p(S), None, Z, T, 1, _p(P), _st()), "softmax_fwd")
        o = torch.empty((nimg, C), dtype=_F32, device=dev)
        gemm(P, Vt, o, 1, hd, T, a=(0, 1), b=(C, 1), c=(hd, 1), batch=Z, inner=heads, a_bs=(heads * T, T),
             b_bs=(T * C, hd), c_bs=(C, hd))
        ctx.save_for_backward(tok, wq, wk, wv, Kt, Vt, q, P)
        ctx.cfg = (nimg, T, C, heads, hd, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        tok, wq, wk, wv, Kt, Vt, q, P = ctx.saved_tensors
        nimg, T, C, heads, hd, scale = ctx.cfg
        do = _c(do)
        dev = do.device
        Z = nimg * heads
        tok2 = tok.view(nimg * T, C)
        # dP[img,h,t] = do[img,h,:] . Vt[img,t,h,:] ;  dVt[img,t,h,:] = P[img,h,t] do[img,h,:]
        dP = torch.empty_like(P)
        gemm(do, Vt, dP, 1, T, hd, a=(0, 1), b=(1, C), c=(T, 1), batch=Z, inner=heads, a_bs=(C, hd), b_bs=(T * C, hd),
             c_bs=(heads * T, T))
        dVt = torch.empty_like(Vt)
        gemm(P, do, dVt, T, hd, 1, a=(1, 0), b=(0, 1), c=(C, 1), batch=Z, inner=heads, a_bs=(heads * T, T), b_bs=(C, hd),
             c_bs=(T * C, hd))
        dS = torch.empty_like(P)
        _ck(_abi.lib().ptx_op_softmax_bwd(_p(P), _p(dP), None, Z, T, 1, _p(dS), _st()), "softmax_bwd")
        dq = torch.empty_like(q)
        gemm(dS, Kt, dq, 1, hd, T, a=(0, 1), b=(C, 1), c=(hd, 1), batch=Z, inner=heads, a_bs=(heads * T, T),
             b_bs=(T * C, hd), c_bs=(C, hd), alpha=scale)
        dKt = torch.empty_like(Kt)
        gemm(dS, q, dKt, T, hd, 1, a=(1, 0), b=(0, 1), c=(C, 1), batch=Z, inner=heads, a_bs=(heads * T, T), b_bs=(C, hd),
             c_bs=(T * C, hd), alpha=scale)
        dtok = linear_nt(dKt, transpose2d(wk))
        if _tuned_ok(dVt, C):
            wvT = transpose2d(wv)
            d2 = torch.empty_like(dtok)
            _ck(_abi.lib().ptx_linear(_p(dVt), _p(wvT), None, _p(dtok), _p(d2), nimg * T, C, C, 0, _st()), "ptx_linear")
            dtok = d2
        else:
            mm(dVt, wv, out=dtok, accumulate=True)
        gemm(dq, wq, dtok, nimg, C, C, a=(C, 1), b=(C, 1), c=(T * C, 1), accumulate=True)      # token-0 rows
        dwk, dbk = mm(dKt, tok2, ta=True), colsum(dKt)
        dwv, dbv = mm(dVt, tok2, ta=True), colsum(dVt)
        dwq = torch.empty_like(wq)
        gemm(dq, tok, dwq, C, C, nimg, a=(1, C), b=(T * C, 1), c=(C, 1))
        return dtok.view(nimg, T, C), dwq, colsum(dq), dwk, dbk, dwv, dbv, None


class _ImgPool(torch.autograd.Function):
    """channel_mapper + AttentionPool2d as ONE node on the folded form of csrc/train_img.hip: the pixel tokens, their keys and
    values are never materialised (only token 0 queries).  Without the tail arguments the node ends at the attention output (before
    c_proj); with ``cw, cb, lnw, lnb`` the same two calls also run c_proj and norm_img (PRE:177, 450) and the node returns the
    image proxies."""

    @staticmethod
    def forward(ctx, img, wc, bc, pos, wq, bq, wk, bk, wv, bv, heads, cw=None, cb=None, lnw=None, lnb=None, eps=1e-5):
        lib = _abi.lib()
        img = _c(img)
        nimg, Cin, hw = img.shape
        C = wc.shape[0]
        tail = cw is not None
        a = _abi.PtxTrainImgPool()
        a.nimg, a.Cin, a.hw, a.C, a.heads = nimg, Cin, hw, C, heads
        a.img_dtype = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[img.dtype]
        params = tuple(_c(t) for t in (wc.reshape(C, Cin), bc, pos, wq, bq, wk, bk, wv, bv) + ((cw, cb, lnw, lnb) if tail else ()))
        a.img = _p(img)
        a.wc, a.bc, a.pos, a.wq, a.bq, a.wk, a.bk, a.wv, a.bv = (_p(t) for t in params[:9])
        if tail:
            a.cw, a.cb, a.lnw, a.lnb = (_p(t) for t in params[9:])
            a.ln_eps = eps
        s0, s1, s2 = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        _ck(lib.ptx_train_imgpool_sizes(ctypes.byref(a), ctypes.byref(s0), ctypes.byref(s1), ctypes.byref(s2)), "ptx_train_imgpool_sizes")
        dev = img.device
        save = torch.empty((s0.value,), dtype=_F32, device=dev)
        tmp = torch.empty((s1.value,), dtype=_F32, device=dev)
        o = torch.empty((nimg, C), dtype=_F32, device=dev)
        if tail:
            a.proxy = _p(o)
        else:
            a.o = _p(o)
        a.save, a.save_floats, a.tmp, a.tmp_floats = _p(save), s0.value, _p(tmp), s1.value
        _ck(lib.ptx_train_imgpool_fwd(ctypes.byref(a), _st()), "ptx_train_imgpool_fwd")
        ctx.save_for_backward(img, save, *params)
        ctx.args, ctx.bwd_floats, ctx.wc_shape, ctx.tail = a, s2.value, wc.shape, tail
        return o

    @staticmethod
    def backward(ctx, do):
        lib = _abi.lib()
        img, save, *params = ctx.saved_tensors
        a = ctx.args
        do = _c(do)
        dev = do.device
        grads = _flat_grads(params, dev)
        dimg = torch.empty_like(img) if ctx.needs_input_grad[0] else None
        tmp = torch.empty((ctx.bwd_floats,), dtype=_F32, device=dev)
        a.dimg, a.tmp, a.tmp_floats = _p(dimg), _p(tmp), ctx.bwd_floats
        a.dwc, a.dbc, a.dpos, a.dwq, a.dbq, a.dwk, a.dbk, a.dwv, a.dbv = (_p(g) for g in grads[:9])
        if ctx.tail:
            a.dproxy = _p(do)
            a.dcw, a.dcb, a.dlnw, a.dlnb = (_p(g) for g in grads[9:])
        else:
            a.dout = _p(do)
        _ck(lib.ptx_train_imgpool_bwd(ctypes.byref(a), _st()), "ptx_train_imgpool_bwd")
        grads[0] = grads[0].view(ctx.wc_shape)
        if not ctx.tail:
            grads = grads + [None] * 4
        return (dimg, *grads[:9], None, *grads[9:], None)


def _imgpool_ok(img3, C, heads):
    nimg, Cin, hw = img3 if isinstance(img3, tuple) else img3.shape
    return _FUSED_IMG and heads == 8 and hw <= 256 and Cin % 8 == 0 and Cin <= 2048 and C % 64 == 0 and C <= 512 and nimg <= 65535


class _AffineApply(torch.autograd.Function):
    """Per-cluster affine + pt_replace + remove_points_by_index (PRE:459-467) through the eval path's compaction
    kernel; returns the per-scene outputs themselves (slices of one buffer), so that the backward takes the B gradients
    as they come instead of autograd scattering each into a zero-filled (B,N,3) tensor first (r04: 30 launches and 0.4 ms
    of host time per step); backward = ptx_op_affine_bwd_list."""

    @staticmethod
    def forward(ctx, kcenter, translate, transform, pts, tag, opos, kidx, kcluster, shape, ws, n_keep):
        lib = _abi.lib()
        B, N = pts.shape[0], pts.shape[1]
        out = torch.empty((B, N, 3), dtype=_F32, device=pts.device)       # only the rows below the counts are returned
        counts = torch.empty((B,), dtype=torch.int32, device=pts.device)
        kcenter, translate, transform = _c(kcenter), _c(translate), _c(transform)
        _ck(lib.ptx_affine_compact(ctypes.byref(shape), _p(pts), _p(tag), _p(kcenter), _p(translate), _p(transform),
                                   _p(out), _p(counts), _p(ws), ws.numel(), _st()), "ptx_affine_compact")
        ctx.save_for_backward(opos, kidx, kcluster, kcenter, transform)
        ctx.dims = (B, N, shape.Mk, shape.K)
        return tuple(out[b, : n_keep[b]] for b in range(B))

    @staticmethod
    def backward(ctx, *douts):
        opos, kidx, kcluster, kcenter, transform = ctx.saved_tensors
        B, N, Mk, K = ctx.dims
        douts = [None if g is None else _c(g) for g in douts]
        dev = kcenter.device
        dt = torch.empty((B * Mk, 3), dtype=_F32, device=dev)
        dT = torch.empty((B * Mk, 9), dtype=_F32, device=dev)
        dc = torch.empty((B * Mk, 3), dtype=_F32, device=dev)
        ptrs = (ctypes.c_void_p * B)(*[_p(g) for g in douts])
        _ck(_abi.lib().ptx_op_affine_bwd_list(ptrs, _p(opos), _p(kidx), _p(kcluster), _p(kcenter), _p(transform), B, N, Mk, K,
                                              _p(dt), _p(dT), _p(dc), _st()), "ptx_op_affine_bwd_list")
        return dc, dt, dT, None, None, None, None, None, None, None, None


# --------------------------------------------------------------------------- the train-mode forward
_SITES_PER_BLOCK = 8         # seed slots of one ProxyBlock: 0, 1 attention (query_attn, proxy-as-key attn), 2..6 the rest


def site_seeds(torch_seed: int, call: int, salt: int = 0):
    """Dropout seeds of one training call: ``seeds[branch][i]`` for the six dropout / DropPath sites of ``_block``.
    ``k_dropout`` is a pure function of (seed, element), so every site needs a seed of its own: the attention node
    draws its two masks with ``seeds[b][0]`` and ``seeds[b][0] + 1``, hence the other five sites start at +2
    (``_SITES_PER_BLOCK`` slots per branch; ``salt`` separates module instances that share torch's seed)."""
    base = (torch_seed * 1000003 + call * 7919 + salt * 104729) & 0x7FFFFFFFFFFF
    return [[base + _SITES_PER_BLOCK * b + (0 if i == 0 else i + 1) for i in range(6)] for b in range(2)]


def _flat_grads(params, dev, slots=None):
    """One allocation for the gradients of ``params`` (None entries stay None): a tuple of tensors shaped like the parameters,
    each starting on a 256-byte boundary; ``slots[i]`` (a ctypes pointer array) receives the addresses.  One split + one view per
    multi-dimensional parameter instead of a slice and a view each (r04: 110 tensor operations per step for two blocks)."""
    sizes = [0 if t is None else (t.numel() + 63) // 64 * 64 for t in params]
    flat = torch.empty((sum(sizes),), dtype=_F32, device=dev)
    base = flat.data_ptr()
    parts = flat.split_with_sizes(sizes)
    grads, off = [], 0
    for i, t in enumerate(params):
        if t is None:
            grads.append(None)
            if slots is not None:
                slots[i] = None
            continue
        g, n = parts[i], t.numel()
        if sizes[i] != n:
            g = g[:n]
        if t.dim() != 1:
            g = g.view(t.shape)
        grads.append(g)
        if slots is not None:
            slots[i] = base + 4 * off
        off += sizes[i]
    return grads


class _BlockFused(torch.autograd.Function):
    """ProxyBlock + trailing LayerNorm + Linear head + BatchNorm1d as ONE node: forward and backward are one C call each
    (ptx_train_block_fwd / _bwd, csrc/train_fused.hip) that enqueue every kernel of the chain from C++."""

    @staticmethod
    def forward(ctx, x, proxy, mask, cfg, run_mean, run_var, *params):
        lib = _abi.lib()
        x, proxy = _c(x), _c(proxy)
        a = _abi.PtxTrainBlock()
        (a.B, a.n, a.L, a.C, a.H, a.heads, a.s, a.nout, a.eps1, a.eps2, a.eps3, a.bn_eps, a.bn_momentum, a.p_attn, a.p_drop,
         a.p_path, seeds) = cfg[:17]
        a.compute_dtype = cfg[17] if len(cfg) > 17 else 0
        for i, sd in enumerate(seeds):
            a.seed[i] = sd & 0xFFFFFFFFFFFFFFFF
        a.x, a.proxy, a.mask = _p(x), _p(proxy), _p(mask)
        params = tuple(None if t is None else _c(t) for t in params)
        for i, t in enumerate(params):
            a.param[i] = _p(t)
        a.bn_run_mean, a.bn_run_var = _p(run_mean), _p(run_var)
        s0, s1, s2 = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        _ck(lib.ptx_train_block_sizes(ctypes.byref(a), ctypes.byref(s0), ctypes.byref(s1), ctypes.byref(s2)), "ptx_train_block_sizes")
        sz = (s0.value, s1.value, s2.value)
        dev = x.device
        save = torch.empty((sz[0],), dtype=_F32, device=dev)
        tmp = torch.empty((sz[1],), dtype=_F32, device=dev)
        out = torch.empty((a.B * a.n, a.nout), dtype=_F32, device=dev)
        a.out, a.save, a.save_floats, a.tmp, a.tmp_floats = _p(out), _p(save), sz[0], _p(tmp), sz[1]
        _ck(lib.ptx_train_block_fwd(ctypes.byref(a), _st()), "ptx_train_block_fwd")
        ctx.save_for_backward(x, proxy, save, *[t for t in params if t is not None])
        ctx.present = [t is not None for t in params]
        ctx.mask = mask
        ctx.args, ctx.bwd_floats = a, sz[2]
        return out

    @staticmethod
    def backward(ctx, dout, dx_add=None):
        # dx_add (one-node step only): a gradient of the same rows that is added into dx by the last kernel of this backward
        lib = _abi.lib()
        saved = ctx.saved_tensors
        x, proxy, save = saved[0], saved[1], saved[2]
        params, k = [], 3
        for pres in ctx.present:
            params.append(saved[k] if pres else None)
            k += 1 if pres else 0
        a = ctx.args
        dout = _c(dout)
        dev = dout.device
        grads = _flat_grads(params, dev, a.grad)                            # every parameter gradient, one allocation
        dx, dproxy = torch.empty_like(x), torch.empty_like(proxy)
        tmp = torch.empty((ctx.bwd_floats,), dtype=_F32, device=dev)
        a.dout, a.dx, a.dproxy, a.tmp, a.tmp_floats = _p(dout), _p(dx), _p(dproxy), _p(tmp), ctx.bwd_floats
        a.dx_add = _p(dx_add)
        _ck(lib.ptx_train_block_bwd(ctypes.byref(a), _st()), "ptx_train_block_bwd")
        return (dx, dproxy, None, None, None, None, *grads)


def _block_fused_ok(mod, blk, head, n, L):
    lib = _abi.lib()
    return (_FUSED_BLOCK and lib.ptx_train_attn_tmp_floats(1, n, L, mod.num_heads, mod.embed_dim) > 0 and head.weight.shape[0] <= 9
            and mod.embed_dim <= 512 and blk.mlp.fc1.weight.shape[0] <= 2048)


def _block_params(blk, out_norm, head, head_bn):
    """The parameters of one fused block call in PTX_TB_* order."""
    a = blk.attn
    return (blk.norm1.weight, blk.norm1.bias, a.pb_bias, a.pc_bias, a.pr_bias, a.qkv.weight, a.qkv.bias, a.proxy_proj.weight,
            a.proxy_proj.bias, a.proj.weight, a.proj.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
            blk.mlp.fc2.weight, blk.mlp.fc2.bias, out_norm.weight, out_norm.bias, head.weight, head.bias, head_bn.weight,
            head_bn.bias)


def _block_cfg(mod, blk, out_norm, head, head_bn, B, n, L, seeds, mods=None, params=None):
    """(cfg tuple, parameters in PTX_TB_* order) of one fused block call.  ``mods`` = (norm1, norm2) and ``params`` come from the
    module's per-weights cache in the one-node step (_static); the scalars are read from their owners on every call."""
    if params is None:
        params = _block_params(blk, out_norm, head, head_bn)
    n1, n2 = mods if mods is not None else (blk.norm1, blk.norm2)
    cfg = (B, n, L, mod.embed_dim, params[13].shape[0], mod.num_heads, params[3].shape[2], params[19].shape[0],
           n1.eps, n2.eps, out_norm.eps, head_bn.eps, head_bn.momentum, float(mod.attn_drop_rate),
           float(mod.drop_rate), float(mod._dpr_last(blk)), tuple(seeds), 1 if getattr(mod, "compute_dtype", "fp32") == "bf16" else 0)
    return cfg, params


def _static(mod):
    """Sub-modules and parameter tuples of the one-node step, looked up once per weights generation (nn.Module.__getattr__ is
    slow: ~250 look-ups per step were 3 % of a host-bound step).  Dropped together with the live-parameter list whenever a
    Parameter / buffer / sub-module OBJECT changes (module._run_train), so swapped objects are seen; scalars (eps, momentum,
    drop rates) are NOT cached."""
    s = getattr(mod, "_train_static", None)
    if s is None:
        off, enc, ap = mod.get_deformable_cluster.get_offsets, mod.simple_encoder, mod.attn_pool2d
        bn, ebn = off.mlp[1], enc.mlp[1]
        tb, ib = mod.textformer[-1], mod.imgformer[-1]
        tn, inn, tt, it, ttn, itn = mod.text_norm[-1], mod.img_norm[-1], mod.text_trans, mod.img_trans, mod.text_trans_norm, mod.img_trans_norm
        s = mod._train_static = dict(
            off=off, bn=bn, enc=enc, ebn=ebn, ap=ap, norm_img=mod.norm_img,
            off_par=(off.mlp[0].weight, off.mlp[0].bias, bn.weight, bn.bias), off_run=(bn.running_mean, bn.running_var),
            oh=off.channel_mapper.weight,
            enc_par=(enc.mlp[0].weight, enc.mlp[0].bias, ebn.weight, ebn.bias), enc_run=(ebn.running_mean, ebn.running_var),
            ip_par=(mod.channel_mapper.weight, mod.channel_mapper.bias, ap.positional_embedding, ap.q_proj.weight, ap.q_proj.bias,
                    ap.k_proj.weight, ap.k_proj.bias, ap.v_proj.weight, ap.v_proj.bias, ap.c_proj.weight, ap.c_proj.bias,
                    mod.norm_img.weight, mod.norm_img.bias),
            tb=(tb, tn, tt, ttn, (tb.norm1, tb.norm2), _block_params(tb, tn, tt, ttn), (ttn.running_mean, ttn.running_var)),
            ib=(ib, inn, it, itn, (ib.norm1, ib.norm2), _block_params(ib, inn, it, itn), (itn.running_mean, itn.running_var)))
    return s


def _block(mod, blk, out_norm, head, head_bn, xa, xb, proxy2d, mask_u8, B, n, L, seeds):
    """One ProxyBlock in train mode (PRE:273-276) + trailing LayerNorm + Linear head + BatchNorm1d (PRE:441-446)."""
    C, heads = mod.embed_dim, mod.num_heads
    a = blk.attn
    if xa is xb and _block_fused_ok(mod, blk, head, n, L):
        cfg, params = _block_cfg(mod, blk, out_norm, head, head_bn, B, n, L, seeds)
        return _BlockFused.apply(xa, proxy2d, mask_u8, cfg, head_bn.running_mean, head_bn.running_var, *params)
    if xa is xb:
        xa, xb = fork(xa, 2)
    s = a.pc_bias.shape[2]
    eps = blk.norm1.eps
    x = _LayerNorm.apply(xa, blk.norm1.weight, blk.norm1.bias, eps)
    x = _SlotBiasAdd.apply(x, a.pb_bias, a.pc_bias, a.pr_bias, n, s)
    qkv = _Linear.apply(x, a.qkv.weight, a.qkv.bias)
    pt = _Linear.apply(proxy2d, a.proxy_proj.weight, a.proxy_proj.bias)
    o = _ProxyAttnCore.apply(qkv, pt, mask_u8, B, n, L, heads, float(mod.attn_drop_rate), seeds[0])
    o = _Linear.apply(o, a.proj.weight, a.proj.bias)
    o = dropout(o, mod.drop_rate, seeds[1])
    o = dropout(o, mod._dpr_last(blk), seeds[2], group=n * C)                      # DropPath: one decision per sample
    x1 = _Add.apply(xb, o)
    x1a, x1b = fork(x1, 2)
    h = _LayerNorm.apply(x1a, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
    h = _Linear.apply(h, blk.mlp.fc1.weight, blk.mlp.fc1.bias)
    h = _Gelu.apply(h)
    h = dropout(h, mod.drop_rate, seeds[3])
    h = _Linear.apply(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias)
    h = dropout(h, mod.drop_rate, seeds[4])
    h = dropout(h, mod._dpr_last(blk), seeds[5], group=n * C)
    x2 = _Add.apply(x1b, h)
    g = _LayerNorm.apply(x2, out_norm.weight, out_norm.bias, out_norm.eps)
    t = _Linear.apply(g, head.weight, head.bias)
    t = _BatchNormRows.apply(t, head_bn.weight, head_bn.bias, head_bn.running_mean, head_bn.running_var, head_bn.eps,
                             head_bn.momentum, False)
    return t


def forward_train(mod, points: List[torch.Tensor], text_feats, text_mask, img_feat, shape, ws, order_override=None):
    """Train-mode forward of ``mod`` (a ProxyTransformationNormReverse).  Returns (list of (N_i',3) tensors with
    grad_fn, dict of index tensors for tests)."""
    lib = _abi.lib()
    dev = points[0].device
    st = _st()
    _tick("enter")
    B, N = len(points), points[0].shape[0]
    M, K, Mt, Mk, C = mod.num_cluster, mod.num_sub, shape.Mt, shape.Mk, mod.embed_dim
    Kd = Mt - Mk
    i32 = dict(dtype=torch.int32, device=dev)
    mod._train_calls += 1
    seeds = site_seeds(torch.initial_seed(), mod._train_calls, mod._instance_salt)
    if _one_node_ok(mod, (B * img_feat.shape[1], mod.input_dim, mod.img_spacial_dim ** 2), Mk, text_feats.shape[1], img_feat.shape[1]):
        st8 = dict(args=(mod, points, text_mask, shape, ws, order_override), seeds=seeds)
        if getattr(mod, "_train_live", None) is None:
            mod._train_live = live_params(mod)              # dropped with the layout check (module.invalidate_weights)
        res = (_TrainStepC if _C_STEP else _TrainStep).apply(st8, text_feats, img_feat, *mod._train_live)
        aux = st8["aux"]
        # the per-cluster transforms are OUTPUTS of the node: a regulariser on module.forward(..., return_transforms=True)'s
        # kcenter / translate / transform reaches the parameters, as it did through the per-operator graph
        n = len(res) - 3
        aux["kcenter"], aux["translate"], aux["transform"] = res[n:]
        return list(res[:n]), aux
    pts = torch.stack([p.detach().to(_F32) for p in points]).contiguous()          # PRE:426-427

    V = img_feat.shape[1]

    def _image_branch():
        # ---- image branch (PRE:449-455), on a side stream: it depends on nothing before it, and the host is ahead of the GPU here
        # (farthest point sampling keeps one work-group per scene busy for ~0.2 ms), so it overlaps the index half and the text
        # block.  Autograd runs a node's backward on the stream of its forward and picks the most recently created ready node
        # first: created HERE -- after the text block -- this chain's backward is enqueued right after the image block's and
        # overlaps the text block's backward (created at the top of the forward it ran last, alone, for 0.5 ms)
        V = img_feat.shape[1]
        hw = mod.img_spacial_dim ** 2
        ap = mod.attn_pool2d
        img3 = _c(img_feat).view(B * V, mod.input_dim, hw)
        main = torch.cuda.current_stream(dev)
        side = _side_stream(mod, dev) if _SIDE_STREAM else None
        if side is not None:
            side.wait_stream(main)
        with torch.cuda.stream(side if side is not None else main):
            if _imgpool_ok(img3, C, mod.num_heads):
                o = _ImgPool.apply(img3, mod.channel_mapper.weight, mod.channel_mapper.bias, ap.positional_embedding, ap.q_proj.weight,
                                   ap.q_proj.bias, ap.k_proj.weight, ap.k_proj.bias, ap.v_proj.weight, ap.v_proj.bias, mod.num_heads)
            else:
                tok = _ImgTokens.apply(img3, mod.channel_mapper.weight, mod.channel_mapper.bias, ap.positional_embedding)
                o = _AttnPoolCore.apply(tok, ap.q_proj.weight, ap.q_proj.bias, ap.k_proj.weight, ap.k_proj.bias, ap.v_proj.weight,
                                        ap.v_proj.bias, mod.num_heads)
            y = _Linear.apply(o, ap.c_proj.weight, ap.c_proj.bias)
            img_proxy = _LayerNorm.apply(y, mod.norm_img.weight, mod.norm_img.bias, mod.norm_img.eps)       # (B*V, C)
        if side is not None:
            img_proxy.record_stream(main)
        return img_proxy, side, main

    if _IMG_POS == 0:
        img_proxy, side, main = _image_branch()
    # ---- index half, part 1: grid centres + ball query #1 (PRE:55-56)
    minmax = torch.empty((B, 2, 3), dtype=_F32, device=dev)
    c0 = torch.empty((B, M, 3), dtype=_F32, device=dev)
    lin = mod._train_lin(dev)
    enc_scratch = torch.empty((max(B * 6, 64),), **i32)        # NOT the lane workspace: its encoded boxes must stay zero
    _ck(lib.ptx_grid_centers(_p(pts), B, N, _p(lin), mod.grid_size, 4.0, _p(minmax), _p(c0), _p(enc_scratch),
                             enc_scratch.numel() * 4, st), "ptx_grid_centers")
    idx1 = torch.empty((B, M, K), **i32)
    cl1 = torch.empty((B, M, K, 3), dtype=_F32, device=dev)
    _ck(lib.ptx_ball_query(_p(c0), _p(pts), B, M, N, K, 3.0, _p(idx1), _p(cl1), None, st), "ptx_ball_query")

    _tick("index1")
    # ---- offset network (PRE:58-62)
    off = mod.get_deformable_cluster.get_offsets
    bn = off.mlp[1]
    pooled = _SlotNet.apply(c0.view(B * M, 3), cl1, off.mlp[0].weight, off.mlp[0].bias, bn.weight, bn.bias,
                            bn.running_mean, bn.running_var, bn.eps, bn.momentum, False)
    centers = _OffsetHead.apply(pooled, off.channel_mapper.weight, c0, minmax, M, 4.0)       # (B*M,3)

    _tick("offset_net")
    # ---- index half, part 2: ball query #2, selection, tags, output positions (PRE:65, 352-420, 478-523)
    cdet = centers.detach()
    if mod._centers_override is not None:
        cdet = mod._centers_override.to(device=dev, dtype=_F32).reshape(B * M, 3).contiguous()
    idx2 = torch.empty((B, M, K), **i32)
    cl2 = torch.empty((B, M, K, 3), dtype=_F32, device=dev)
    pad = torch.empty((B, M), **i32)
    _ck(lib.ptx_ball_query(_p(cdet), _p(pts), B, M, N, K, 3.0, _p(idx2), _p(cl2), _p(pad), st), "ptx_ball_query")
    order = torch.empty((B, Mt), **i32)
    picks = torch.empty((B, max(Kd, 1)), **i32)
    keep = torch.empty((B, Mk), **i32)
    kcenter_i = torch.empty((B, Mk, 3), dtype=_F32, device=dev)
    kcluster = torch.empty((B, Mk, K, 3), dtype=_F32, device=dev)
    kidx = torch.empty((B, Mk, K), **i32)
    drop_idx = torch.empty((B, max(Kd, 1) * K), **i32)
    tag = torch.empty((B, N), **i32)                      # every word is written by ptx_select_clusters
    oo = None if order_override is None else order_override.to(device=dev, dtype=torch.int32).contiguous()
    _ck(lib.ptx_select_clusters(ctypes.byref(shape), _p(idx2), _p(cdet), _p(cl2), _p(pad), _p(oo), _p(order), _p(picks),
                                _p(keep), _p(kcenter_i), _p(kcluster), _p(kidx), _p(drop_idx), _p(tag), st), "ptx_select_clusters")
    if _IMG_POS == 1:
        img_proxy, side, main = _image_branch()
    ntiles = (N + 2047) // 2048
    tile_counts = torch.empty((B * ntiles,), **i32)
    opos = torch.empty((B, N), **i32)
    counts = torch.empty((B,), **i32)
    _ck(lib.ptx_op_out_positions(_p(tag), B, N, _p(tile_counts), _p(opos), _p(counts), st), "ptx_op_out_positions")
    # the list lengths of PRE:467 are known here: copy them out now and wait at the very end, so that the host keeps enqueueing
    # the float half while the index half runs (r04: a blocking read at the end idled the GPU across the forward / backward seam)
    if getattr(mod, "_train_pin", None) is None:
        mod._train_pin = {}
    pin = mod._train_pin.get((B, st))
    if pin is None:
        pin = mod._train_pin[(B, st)] = (torch.empty((B,), dtype=torch.int32, pin_memory=True), torch.cuda.Event())
    pin[1].synchronize()                                         # an earlier call's copy into the same pinned words
    pin[0].copy_(counts, non_blocking=True)
    pin[1].record()
    src = torch.empty((B * Mk,), **i32)
    _ck(lib.ptx_op_keep_rows(_p(order), _p(keep), B, M, Mt, Mk, _p(src), st), "ptx_op_keep_rows")

    _tick("index2")
    # ---- float half: kept centres, point proxies (PRE:437)
    kcenter = _GatherRows.apply(centers, src)                                    # (B*Mk,3), differentiable
    kc_enc, kc_aff = fork(kcenter, 2)
    enc = mod.simple_encoder
    ebn = enc.mlp[1]
    pp = _SlotNet.apply(kc_enc, kcluster, enc.mlp[0].weight, enc.mlp[0].bias, ebn.weight, ebn.bias, ebn.running_mean,
                        ebn.running_var, ebn.eps, ebn.momentum, True)            # (B*Mk, C)
    if _FUSED_BLOCK:                      # the fused block node sums its two uses of the point proxies itself
        pp_t1, pp_i1 = fork(pp, 2)
        pp_t2, pp_i2 = pp_t1, pp_i1
    else:
        pp_t1, pp_t2, pp_i1, pp_i2 = fork(pp, 4)

    _tick("encoder")
    if _IMG_POS == 2:
        img_proxy, side, main = _image_branch()
    # ---- text branch (PRE:440-446)
    L = text_feats.shape[1]
    tf2 = _c(text_feats.to(_F32)).view(B * L, C)
    translate = _block(mod, mod.textformer[-1], mod.text_norm[-1], mod.text_trans, mod.text_trans_norm, pp_t1, pp_t2,
                       tf2, text_mask, B, Mk, L, seeds[0])

    _tick("text_block")
    if _IMG_POS == 3:
        img_proxy, side, main = _image_branch()
    _tick("img_branch")
    if side is not None:
        main.wait_stream(side)
        img_proxy = _StreamHop.apply(img_proxy, side)
    transform = _block(mod, mod.imgformer[-1], mod.img_norm[-1], mod.img_trans, mod.img_trans_norm, pp_i1, pp_i2,
                       img_proxy, None, B, Mk, V, seeds[1])

    _tick("img_block")
    # ---- submanifold reshape + scatter + drop (PRE:459-467)
    pin[1].synchronize()
    n_keep = pin[0].tolist()                                                    # the list lengths of PRE:467
    outs = list(_AffineApply.apply(kc_aff, translate, transform, pts, tag, opos, kidx, kcluster, shape, ws, n_keep))
    _tick("affine")
    aux = dict(idx2=idx2, order=order, picks=picks[:, :Kd], keep=keep, kidx=kidx, drop_idx=drop_idx[:, : Kd * K],
               centers=centers, translate=translate, transform=transform, point_proxy=pp, img_proxy=img_proxy,
               kcenter=kcenter, opos=opos)
    return outs, aux


# --------------------------------------------------------------------------- the whole float half as ONE autograd node
# r04: with every stage fused the step was host-bound again, and a third of the host time was autograd itself: ~12 custom
# Function.apply calls forward, as many engine callbacks backward, ~60 AccumulateGrad hand-overs, stream bookkeeping per node.
# The graph of this path is static, so the training forward below runs the SAME node bodies (their static forward / backward
# methods, on a plain context object) in program order inside one Function, and its backward walks them in reverse with the
# gradient sums written out.  The per-operator nodes above remain the unit-tested building blocks and the fallback for shapes
# outside the fused kernels' range.
class _Ctx:
    """What a node body expects of its ``ctx``."""

    def __init__(self, needs=()):
        self.needs_input_grad = needs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _one_node_ok(mod, img3, n, L, V):
    return (_ONE_NODE and _imgpool_ok(img3, mod.embed_dim, mod.num_heads) and
            _block_fused_ok(mod, mod.textformer[-1], mod.text_trans, n, L) and _block_fused_ok(mod, mod.imgformer[-1], mod.img_trans, n, V))


class _TrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, st8, text_feats, img_feat, *params):
        mod, points, text_mask, shape, ws, order_override = st8["args"]
        lib = _abi.lib()
        dev = points[0].device
        st = _st()
        B, N = len(points), points[0].shape[0]
        M, K, Mt, Mk, C = mod.num_cluster, mod.num_sub, shape.Mt, shape.Mk, mod.embed_dim
        Kd = Mt - Mk
        i32 = dict(dtype=torch.int32, device=dev)
        # PRE:426-427 (no graph is recorded inside a Function's forward: no detach; _check_inputs has already looked at the dtypes)
        pts = torch.stack(points) if all([p.dtype == _F32 for p in points]) else torch.stack([p.to(_F32) for p in points])
        seeds = st8["seeds"]
        T = {}                                                        # the tape: one context per node body
        # ---- image branch on the side stream (PRE:449-450): it needs nothing from the clustering half, so it is enqueued FIRST and runs
        # beside the ball queries and the farthest point sampling (one work-group per scene for 0.2 ms: the chip is idle next to it)
        V = img_feat.shape[1]
        hw = mod.img_spacial_dim ** 2
        S = _static(mod)
        ip_par = S["ip_par"]
        img3 = _c(img_feat).view(B * V, mod.input_dim, hw)
        main = torch.cuda.current_stream(dev)
        side = _side_stream(mod, dev) if _SIDE_STREAM else None

        def run_img():
            if side is not None:
                side.wait_stream(main)
            with torch.cuda.stream(side if side is not None else main):
                T["ip"] = _Ctx((ctx.needs_input_grad[2],))
                return _ImgPool.forward(T["ip"], img3, *ip_par[:9], mod.num_heads, *ip_par[9:], S["norm_img"].eps)
        img_proxy = run_img() if _IMG_FIRST else None
        # ---- index half, part 1 + offset network (PRE:55-62)
        minmax = torch.empty((B, 2, 3), dtype=_F32, device=dev)
        c0 = torch.empty((B, M, 3), dtype=_F32, device=dev)
        lin = mod._train_lin(dev)
        enc_scratch = torch.empty((max(B * 6, 64),), **i32)
        _ck(lib.ptx_grid_centers(_p(pts), B, N, _p(lin), mod.grid_size, 4.0, _p(minmax), _p(c0), _p(enc_scratch),
                                 enc_scratch.numel() * 4, st), "ptx_grid_centers")
        idx1 = torch.empty((B, M, K), **i32)
        cl1 = torch.empty((B, M, K, 3), dtype=_F32, device=dev)
        _ck(lib.ptx_ball_query(_p(c0), _p(pts), B, M, N, K, 3.0, _p(idx1), _p(cl1), None, st), "ptx_ball_query")
        bn = S["bn"]
        T["off"] = _Ctx((False,) * 11)
        pooled = _SlotNet.forward(T["off"], c0.view(B * M, 3), cl1, *S["off_par"], *S["off_run"], bn.eps, bn.momentum, False)
        T["oh"] = _Ctx()
        centers = _OffsetHead.forward(T["oh"], pooled, S["oh"], c0, minmax, M, 4.0)
        # ---- index half, part 2 (PRE:65, 352-420, 478-523)
        cdet = centers
        if mod._centers_override is not None:
            cdet = mod._centers_override.to(device=dev, dtype=_F32).reshape(B * M, 3).contiguous()
        idx2 = torch.empty((B, M, K), **i32)
        cl2 = torch.empty((B, M, K, 3), dtype=_F32, device=dev)
        pad = torch.empty((B, M), **i32)
        _ck(lib.ptx_ball_query(_p(cdet), _p(pts), B, M, N, K, 3.0, _p(idx2), _p(cl2), _p(pad), st), "ptx_ball_query")
        order = torch.empty((B, Mt), **i32)
        picks = torch.empty((B, max(Kd, 1)), **i32)
        keep = torch.empty((B, Mk), **i32)
        kcenter_i = torch.empty((B, Mk, 3), dtype=_F32, device=dev)
        kcluster = torch.empty((B, Mk, K, 3), dtype=_F32, device=dev)
        kidx = torch.empty((B, Mk, K), **i32)
        drop_idx = torch.empty((B, max(Kd, 1) * K), **i32)
        tag = torch.empty((B, N), **i32)                      # every word is written by ptx_select_clusters
        oo = None if order_override is None else order_override.to(device=dev, dtype=torch.int32).contiguous()
        _ck(lib.ptx_select_clusters(ctypes.byref(shape), _p(idx2), _p(cdet), _p(cl2), _p(pad), _p(oo), _p(order), _p(picks),
                                    _p(keep), _p(kcenter_i), _p(kcluster), _p(kidx), _p(drop_idx), _p(tag), st), "ptx_select_clusters")
        if img_proxy is None:
            img_proxy = run_img()
        # ---- output positions; the list lengths of PRE:467 are copied out now and awaited at the very end
        ntiles = (N + 2047) // 2048
        tile_counts = torch.empty((B * ntiles,), **i32)
        opos = torch.empty((B, N), **i32)
        counts = torch.empty((B,), **i32)
        _ck(lib.ptx_op_out_positions(_p(tag), B, N, _p(tile_counts), _p(opos), _p(counts), st), "ptx_op_out_positions")
        if getattr(mod, "_train_pin", None) is None:
            mod._train_pin = {}
        pin = mod._train_pin.get((B, st))
        if pin is None:
            pin = mod._train_pin[(B, st)] = (torch.empty((B,), dtype=torch.int32, pin_memory=True), torch.cuda.Event())
        pin[1].synchronize()
        pin[0].copy_(counts, non_blocking=True)
        pin[1].record()
        src = torch.empty((B * Mk,), **i32)
        _ck(lib.ptx_op_keep_rows(_p(order), _p(keep), B, M, Mt, Mk, _p(src), st), "ptx_op_keep_rows")
        # ---- float half (PRE:437-455)
        T["g"] = _Ctx()
        kcenter = _GatherRows.forward(T["g"], centers, src)
        ebn = S["ebn"]
        T["enc"] = _Ctx((True,) + (False,) * 10)
        pp = _SlotNet.forward(T["enc"], kcenter, kcluster, *S["enc_par"], *S["enc_run"], ebn.eps, ebn.momentum, True)
        L = text_feats.shape[1]
        tf2 = _c(text_feats.to(_F32)).view(B * L, C)
        tbm, ibm = S["tb"], S["ib"]
        cfg_t, par_t = _block_cfg(mod, *tbm[:4], B, Mk, L, seeds[0], tbm[4], tbm[5])
        cfg_i, par_i = _block_cfg(mod, *ibm[:4], B, Mk, V, seeds[1], ibm[4], ibm[5])
        T["tb"], T["ib"] = _Ctx(), _Ctx()
        apart = side is not None and _BLOCKS_APART
        if apart:
            # the image block follows its pooling pass on the side stream, beside the text block on the caller's stream: the two
            # blocks share nothing but their input rows, and half of their kernels are too small to fill the chip alone
            pp.record_stream(side)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                transform = _BlockFused.forward(T["ib"], pp, img_proxy, None, cfg_i, *ibm[6], *par_i)
            transform.record_stream(main)
        translate = _BlockFused.forward(T["tb"], pp, tf2, text_mask, cfg_t, *tbm[6], *par_t)
        if side is not None:
            img_proxy.record_stream(main)
            main.wait_stream(side)
        if not apart:
            transform = _BlockFused.forward(T["ib"], pp, img_proxy, None, cfg_i, *ibm[6], *par_i)
        # ---- submanifold reshape + scatter + drop (PRE:459-467)
        pin[1].synchronize()
        n_keep = pin[0].tolist()
        T["aff"] = _Ctx()
        outs = _AffineApply.forward(T["aff"], kcenter, translate, transform, pts, tag, opos, kidx, kcluster, shape, ws, n_keep)
        st8["aux"] = dict(idx2=idx2, order=order, picks=picks[:, :Kd], keep=keep, kidx=kidx, drop_idx=drop_idx[:, : Kd * K],
                          centers=centers, translate=translate, transform=transform, point_proxy=pp, img_proxy=img_proxy,
                          kcenter=kcenter, opos=opos)
        ctx.tape = T
        ctx.set_materialize_grads(False)                # unused outputs (normally the three transforms) arrive as None, not as zero fills
        ctx.out_like = [(o.shape, o.dtype) for o in outs]
        ctx.streams = (main, side)
        if S.get("pids_of") is not mod._train_live or len(S["pids"]) != len(params):     # `params` IS mod._train_live (forward_train)
            S["pids_of"], S["pids"] = mod._train_live, [id(p) for p in params]
        ctx.meta = (text_feats.shape, text_feats.dtype, img_feat.shape, S["pids"],
                    dict(off=S["off_par"], oh=S["oh"], enc=S["enc_par"], tb=par_t, ib=par_i, ip=ip_par))
        # fresh aliases, NOT the objects the tape holds: a returned tensor gets this node as its grad_fn, and the node owns ctx ->
        # tape -> that tensor -- a cycle through C++ that Python's collector cannot see (r05: every step's activations, 190 MiB at
        # the training shape, stayed allocated for good)
        return (*outs, kcenter.detach(), translate.detach(), transform.detach())

    @staticmethod
    def backward(ctx, *douts):
        T = ctx.tape
        main, side = ctx.streams
        tf_shape, tf_dtype, img_shape, pids, P = ctx.meta
        G = {}                                               # id(parameter) -> gradient

        def put(params, grads):
            for p_, g_ in zip(params, grads):
                if p_ is not None and g_ is not None:
                    G[id(p_)] = g_

        n = len(ctx.out_like)
        g_kc, g_tr, g_tf = douts[n:n + 3]               # gradients that arrive through return_transforms' tensors (None: unused)
        dev = T["aff"].saved_tensors[3].device
        douts = [g if g is not None else torch.zeros(shp, dtype=dt_, device=dev) for g, (shp, dt_) in zip(douts[:n], ctx.out_like)]
        dkc_aff, dtranslate, dtransform = _AffineApply.backward(T["aff"], *douts)[:3]
        if g_kc is not None:
            dkc_aff = add_(_c(dkc_aff), _c(g_kc.to(_F32)).view(dkc_aff.shape))
        if g_tr is not None:
            dtranslate = add_(_c(dtranslate), _c(g_tr.to(_F32)).view(dtranslate.shape))
        if g_tf is not None:
            dtransform = add_(_c(dtransform), _c(g_tf.to(_F32)).view(dtransform.shape))
        apart = side is not None and _BLOCKS_APART
        side_grads = []
        if apart:
            # image block AND image branch on the side stream, beside the text block's backward on the caller's stream
            dtransform = _c(dtransform)
            dtransform.record_stream(side)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                r = _BlockFused.backward(T["ib"], dtransform)
                dpp_i, dproxy_i = r[0], r[1]
                put(P["ib"], r[6:])
                side_grads += [g for g in r[6:] if g is not None][:1]      # views of ONE allocation (_flat_grads): one record covers it
                dpp_ready = torch.cuda.Event()
                dpp_ready.record(side)
        else:
            r = _BlockFused.backward(T["ib"], dtransform)
            dpp_i, dproxy_i = r[0], r[1]
            put(P["ib"], r[6:])
            # image branch on its side stream, overlapping the text block's backward below
            if side is not None:
                dproxy_i.record_stream(side)
                side.wait_stream(main)
        with torch.cuda.stream(side if side is not None else main):
            r3 = _ImgPool.backward(T["ip"], dproxy_i)          # (dimg, 9 grads, None, 4 tail grads, None)
            put(P["ip"], list(r3[1:10]) + list(r3[11:15]))
            dimg = r3[0]
            side_grads += [g for g in r3 if g is not None]
        if apart:
            r = _BlockFused.backward(T["tb"], dtranslate)
            dpp_i.record_stream(main)
            main.wait_event(dpp_ready)
            dpp, dtf2 = add_(r[0], dpp_i), r[1]
        else:
            r = _BlockFused.backward(T["tb"], dtranslate, dpp_i)       # dx = both blocks' gradients of the point proxies
            dpp, dtf2 = r[0], r[1]
        put(P["tb"], r[6:])
        r = _SlotNet.backward(T["enc"], dpp)
        put(P["enc"], r[2:6])
        dkcenter = add_(r[0], _c(dkc_aff))
        dcenters = _GatherRows.backward(T["g"], dkcenter)[0]
        r = _OffsetHead.backward(T["oh"], dcenters)
        G[id(P["oh"])] = r[1]
        r = _SlotNet.backward(T["off"], r[0])
        put(P["off"], r[2:6])
        if side is not None:
            for g in side_grads:
                g.record_stream(main)
            main.wait_stream(side)
        dtext = dtf2.view(tf_shape).to(tf_dtype) if ctx.needs_input_grad[1] else None
        dimg = dimg.view(img_shape) if (dimg is not None and ctx.needs_input_grad[2]) else None
        return (None, dtext, dimg, *[G.get(i) for i in pids])


# --------------------------------------------------------------------------- the one node with its bodies in C++ (r06)
# _TrainStep above runs ~35 library calls and ~60 allocations per step from Python; its host time (0.6 ms of Python around 0.5 ms
# inside the library) was what a box with a slow host ran the step at.  Here the node keeps the autograd plumbing and the
# ALLOCATIONS -- one arena per direction, the output buffer, one gradient buffer -- and each direction is ONE call: the library
# carves every intermediate out of the arenas (ptx_train_step_layout names the offsets) and enqueues the same launches in the same
# order on the same two streams.  Same kernels, same bits: tests/test_gpu_train.py holds both forms to the same fixtures.
class _Aux(dict):
    """The step's intermediates as views of the forward arena, made on first access (tests / return_transforms read a few)."""

    def __init__(self, arena, spec):
        super().__init__()
        self._arena, self._spec = arena, spec

    def __missing__(self, k):
        off, shape, dt = self._spec[k]
        n = 1
        for d in shape:
            n *= d
        v = self._arena[off:off + n * 4].view(dt).view(shape)
        self[k] = v
        return v

    def __contains__(self, k):
        return k in self._spec or dict.__contains__(self, k)


def _cstep_static(mod, S, dev):
    """Per weights-generation pieces of the C step: the canonical parameter list (PTX_TS_* order), the ctypes struct with its
    constant fields, four events, the pinned count words."""
    # one set per (device, caller's stream): the struct, the events and the pinned count words belong to the steps of ONE stream
    # (a module trained on two streams in turn must not share them -- the Python-bodied node keys its pinned words the same way)
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    table = S.setdefault("cstep", {})
    cs = table.get(key)
    if cs is None:
        off, enc = S["off"], S["enc"]
        bn, ebn = S["bn"], S["ebn"]
        canon = [off.mlp[0].weight, off.mlp[0].bias, bn.weight, bn.bias, S["oh"], enc.mlp[0].weight, enc.mlp[0].bias, ebn.weight,
                 ebn.bias, *S["ip_par"], *S["tb"][5], *S["ib"][5]]
        assert len(canon) == _abi.TS_NGRAD
        st = _abi.PtxTrainStep()
        evs = []
        for _ in range(4):
            e = torch.cuda.Event()
            e.record()                                   # torch creates the hipEvent on first use
            evs.append(e)
        cs = table[key] = dict(dev=str(dev), canon=canon, st=st, ev=evs, layouts={}, pin={},
                               slot_of={id(p_): i for i, p_ in enumerate(canon) if p_ is not None})
        st.ev_fork, st.ev_join, st.ev_pp, st.ev_counts = (e.cuda_event for e in evs)
    return cs


def _cstep_fill(mod, S, cs, shape, pts, text2, text_mask, img3, seeds, ws, order_override, side):
    """Everything of the struct that can change between steps (pointers are re-read every step: a swapped storage must be seen)."""
    st = cs["st"]
    st.shape = shape
    st.points, st.lin = _p(pts), _p(mod._train_lin(pts.device))
    co = mod._centers_override
    keep = []
    if co is not None:
        co = co.to(device=pts.device, dtype=_F32).reshape(-1, 3).contiguous()
        keep.append(co)
    oo = None if order_override is None else order_override.to(device=pts.device, dtype=torch.int32).contiguous()
    keep.append(oo)
    st.centers_override, st.order_override = _p(co), _p(oo)
    st.text_feats, st.text_mask = _p(text2), _p(text_mask)
    bn, ebn = S["bn"], S["ebn"]
    for sn, par, run, b_ in ((st.off, S["off_par"], S["off_run"], bn), (st.enc, S["enc_par"], S["enc_run"], ebn)):
        sn.conv_w, sn.conv_b, sn.bn_w, sn.bn_b = (_p(t) for t in par)
        sn.run_mean, sn.run_var = _p(run[0]), _p(run[1])
        sn.eps, sn.momentum, sn.W = b_.eps, b_.momentum, par[0].shape[0]
    st.map_w = _p(S["oh"])
    ip, ipp = st.ip, S["ip_par"]
    nimg, Cin, hw = img3.shape
    ip.nimg, ip.Cin, ip.hw, ip.C, ip.heads = nimg, Cin, hw, mod.embed_dim, mod.num_heads
    ip.img_dtype = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[img3.dtype]
    ip.img = _p(img3)
    (ip.wc, ip.bc, ip.pos, ip.wq, ip.bq, ip.wk, ip.bk, ip.wv, ip.bv, ip.cw, ip.cb, ip.lnw, ip.lnb) = (_p(t) for t in ipp)
    ip.ln_eps = S["norm_img"].eps
    B, Mk = shape.B, shape.Mk
    for blk, bm, L_, sd, mask in ((st.tb, S["tb"], shape.L, seeds[0], text_mask), (st.ib, S["ib"], shape.V, seeds[1], None)):
        cfg, par = _block_cfg(mod, *bm[:4], B, Mk, L_, sd, bm[4], bm[5])
        (blk.B, blk.n, blk.L, blk.C, blk.H, blk.heads, blk.s, blk.nout, blk.eps1, blk.eps2, blk.eps3, blk.bn_eps, blk.bn_momentum,
         blk.p_attn, blk.p_drop, blk.p_path, sds) = cfg[:17]
        blk.compute_dtype = cfg[17]
        for i, v in enumerate(sds):
            blk.seed[i] = v & 0xFFFFFFFFFFFFFFFF
        for i, t in enumerate(par):
            blk.param[i] = _p(t)
        blk.bn_run_mean, blk.bn_run_var = _p(bm[6][0]), _p(bm[6][1])
    st.ws, st.ws_bytes = _p(ws), ws.numel()
    st.side_stream = None if side is None else side.cuda_stream
    st.blocks_apart = 1 if (side is not None and _BLOCKS_APART) else 0
    return st, keep


class _TrainStepC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, st8, text_feats, img_feat, *params):
        mod, points, text_mask, shape, ws, order_override = st8["args"]
        lib = _abi.lib()
        dev = points[0].device
        B, N = len(points), points[0].shape[0]
        C, Mk = mod.embed_dim, shape.Mk
        pts = torch.stack(points) if all([p.dtype == _F32 for p in points]) else torch.stack([p.to(_F32) for p in points])
        S = _static(mod)
        cs = _cstep_static(mod, S, dev)
        V, L = img_feat.shape[1], text_feats.shape[1]
        img3 = _c(img_feat).view(B * V, mod.input_dim, mod.img_spacial_dim ** 2)
        tf2 = _c(text_feats.to(_F32)).view(B * L, C)
        main = torch.cuda.current_stream(dev)
        side = _side_stream(mod, dev) if _SIDE_STREAM else None
        st, keep = _cstep_fill(mod, S, cs, shape, pts, tf2, text_mask, img3, st8["seeds"], ws, order_override, side)
        lkey = (B, N, L, V, img3.dtype, st.tb.param[_abi.TB_PARAMS.index("qkv_b")] is None)
        lay = cs["layouts"].get(lkey)
        if lay is None:
            lay = _abi.PtxTrainStepLayout()
            _ck(lib.ptx_train_step_layout(ctypes.byref(st), ctypes.byref(lay)), "ptx_train_step_layout")
            cs["layouts"][lkey] = lay
        arena = torch.empty((lay.arena_fwd_bytes,), dtype=torch.uint8, device=dev)
        out = torch.empty((B, N, 3), dtype=_F32, device=dev)
        pin = cs["pin"].get(B)
        if pin is None:
            pin = cs["pin"][B] = torch.empty((B,), dtype=torch.int32, pin_memory=True)
        ev_counts = cs["ev"][3]
        ev_counts.synchronize()                                  # an earlier step's copy into the same pinned words
        st.out, st.counts_host = _p(out), pin.data_ptr()
        st.arena_fwd, st.arena_fwd_bytes = _p(arena), arena.numel()
        if side is not None:
            arena.record_stream(side)
        _ck(lib.ptx_train_step_fwd(ctypes.byref(st), main.cuda_stream), "ptx_train_step_fwd")
        ev_counts.synchronize()                                  # the list lengths of PRE:467
        n_keep = pin.tolist()
        Kd, M, K = shape.Mt - Mk, mod.num_cluster, shape.K
        i32 = torch.int32
        spec = dict(idx2=(lay.idx2, (B, M, K), i32), order=(lay.order, (B, shape.Mt), i32), picks=(lay.picks, (B, Kd), i32),
                    keep=(lay.keep, (B, Mk), i32), kidx=(lay.kidx, (B, Mk, K), i32), drop_idx=(lay.drop_idx, (B, Kd * K), i32),
                    centers=(lay.centers, (B * M, 3), _F32), translate=(lay.translate, (B * Mk, 3), _F32),
                    transform=(lay.transform, (B * Mk, 9), _F32), point_proxy=(lay.point_proxy, (B * Mk, C), _F32),
                    img_proxy=(lay.img_proxy, (B * V, C), _F32), kcenter=(lay.kcenter, (B * Mk, 3), _F32), opos=(lay.opos, (B, N), i32))
        aux = st8["aux"] = _Aux(arena, spec)
        ctx.set_materialize_grads(False)
        ctx.keep = (arena, pts, tf2, img3, text_mask, ws, keep, params)      # everything the backward's raw pointers refer to
        ctx.cs, ctx.lay, ctx.side, ctx.n_out = cs, lay, side, B
        ctx.meta = (text_feats.shape, text_feats.dtype, img_feat.shape, B * L, C)
        ctx.step = (mod, S, shape, st8["seeds"], order_override)
        outs = tuple(out[b, : n_keep[b]] for b in range(B))
        return (*outs, aux["kcenter"].detach(), aux["translate"].detach(), aux["transform"].detach())

    @staticmethod
    def backward(ctx, *douts):
        lib = _abi.lib()
        arena, pts, tf2, img3, text_mask, ws, keep, params = ctx.keep
        cs, lay, side, B = ctx.cs, ctx.lay, ctx.side, ctx.n_out
        mod, S, shape, seeds, order_override = ctx.step
        tf_shape, tf_dtype, img_shape, BL, C = ctx.meta
        dev = arena.device
        main = torch.cuda.current_stream(dev)
        # the struct is shared by every step of this module: fill it again for THIS step (another forward may have run in between)
        st, keep2 = _cstep_fill(mod, S, cs, shape, pts, tf2, text_mask, img3, seeds, ws, order_override, side)
        st.arena_fwd, st.arena_fwd_bytes = _p(arena), arena.numel()
        gs = [None if g is None else _c(g) for g in douts[:B]]
        ptrs = (ctypes.c_void_p * B)(*[_p(g) for g in gs])
        st.douts = ctypes.addressof(ptrs)
        extra = [None if g is None else _c(g.to(_F32)) for g in douts[B:B + 3]]
        st.g_kcenter, st.g_translate, st.g_transform = (_p(g) for g in extra)
        arena_b = torch.empty((lay.arena_bwd_bytes,), dtype=torch.uint8, device=dev)
        grads = torch.empty((lay.grads_floats,), dtype=_F32, device=dev)
        dtext = torch.empty((BL, C), dtype=_F32, device=dev) if ctx.needs_input_grad[1] else None
        dimg = torch.empty_like(img3) if ctx.needs_input_grad[2] else None
        st.arena_bwd, st.arena_bwd_bytes, st.grads, st.grads_floats = _p(arena_b), arena_b.numel(), _p(grads), grads.numel()
        st.dtext, st.dimg = _p(dtext), _p(dimg)
        if side is not None:
            for t in (arena_b, grads, dimg):
                if t is not None:
                    t.record_stream(side)
        _ck(lib.ptx_train_step_bwd(ctypes.byref(st), main.cuda_stream), "ptx_train_step_bwd")
        slot_of, canon, goff = cs["slot_of"], cs["canon"], lay.grad_off
        out = []
        for p_ in params:
            i = slot_of.get(id(p_))
            if i is None or goff[i] < 0:
                out.append(None)
                continue
            n = p_.numel()
            g = grads[goff[i]:goff[i] + n]
            out.append(g if p_.dim() == 1 else g.view(p_.shape))
        dt_ = None if dtext is None else dtext.view(tf_shape).to(tf_dtype)
        di_ = None if dimg is None else dimg.view(img_shape)
        return (None, dt_, di_, *out)


def live_params(mod):
    """The parameters that receive a gradient (the last block of each list and everything outside the block lists: SURVEY H8)."""
    off, enc, ap = mod.get_deformable_cluster.get_offsets, mod.simple_encoder, mod.attn_pool2d
    mods = [off, enc, mod.channel_mapper, ap, mod.norm_img, mod.textformer[-1], mod.text_norm[-1], mod.text_trans, mod.text_trans_norm,
            mod.imgformer[-1], mod.img_norm[-1], mod.img_trans, mod.img_trans_norm]
    out, seen = [], set()
    for m_ in mods:
        for p_ in m_.parameters():
            if id(p_) not in seen:
                seen.add(id(p_)); out.append(p_)
    return out
