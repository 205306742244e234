"""Multi-view depth ingest on the device: the part of the reference's data pipeline
between the decoded depth maps and the ``(N,3)`` cloud that ``ProxyTransformationNormReverse.forward`` consumes
(configs/grounding/proxy-tiblock33-gs12-wbias-ddr0.6-clip.py:105-142):

    ConvertRGBDToPoints(coord_type='CAMERA')        datasets/transforms/points.py:20-98
    PointSample(num_points=n_points // 10)          datasets/transforms/points.py:290-420      per view
    AggregateMultiViewPoints(coord_type='DEPTH')    datasets/transforms/multiview.py:195-253
    PointSample(num_points=n_points)                datasets/transforms/points.py:290-420      the scene
    GlobalRotScaleTrans (points only, optional)     datasets/transforms/augmentation.py:253-   train pipeline

The random draws stay where the reference has them -- ``np.random.choice`` on the host, consumed in the reference's order
(one draw per non-empty view in view order, then one for the scene) so that a seeded run picks the same pixels -- and
are composed into ONE index per output point.  Everything else runs in HIP behind the C ABI (``ptx_ingest_index`` /
``ptx_ingest_gather``, csrc/ingest.hip): a streaming pass over the depth maps builds a rank / select index of the pixels
with depth != 0, then only the N selected points are un-projected, moved to the global frame and written, together with
the cloud's bounding box in the encoding the forward's clustering kernel reads (``forward(..., bbox=batch.bbox)`` skips
its min / max pass, PRE:37-38).  There is no CPU path: tensors must be on the GPU and the library must be built.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import threading

import numpy as np
import torch

from . import _abi

__all__ = ["MultiViewIngest", "IngestedBatch", "compose_choices", "lu_factor_4x4", "lu_factor_4x4_batch"]

_DEPTH_DTYPES = {torch.float32: 0, torch.uint16: 1, torch.int16: 1}       # int16: a reinterpreted uint16 image


@dataclass
class IngestedBatch:
    points: List[torch.Tensor]          # B views (N,3) of one (B,N,3) buffer: the ``points`` argument of the forward
    bbox: torch.Tensor                  # (B,6) int32 = encoded min / max per scene (``forward(..., bbox=...)``)
    view_counts: List[np.ndarray]       # per scene: pixels with depth != 0 per view (the reference's len(points))
    sel: List[np.ndarray]               # per scene: the composed index per output point (tests / debugging)


def lu_factor_4x4(a: np.ndarray):
    """LU factors of a 4x4 float32 matrix with partial pivoting (what ``torch.linalg.solve`` factors internally,
    multiview.py:232): returns (lu, rows) with ``a[rows] = L @ U``, L unit lower and U upper packed into ``lu``."""
    a = np.array(a, dtype=np.float32).reshape(4, 4).copy()
    rows = [0, 1, 2, 3]
    for k in range(4):
        p = k + int(np.argmax(np.abs(a[k:, k])))
        if a[p, k] == 0.0:
            raise ValueError("singular global2ego matrix")
        if p != k:
            a[[k, p]] = a[[p, k]]
            rows[k], rows[p] = rows[p], rows[k]
        for i in range(k + 1, 4):
            a[i, k] = np.float32(a[i, k] / a[k, k])
            a[i, k + 1:] = a[i, k + 1:] - a[i, k] * a[k, k + 1:]
    return a, np.asarray(rows, dtype=np.int32)


def lu_factor_4x4_batch(a: np.ndarray):
    """``lu_factor_4x4`` of V matrices at once, the same float32 operations in the same order (vectorised over the views: 50 views
    x 6 scenes of Python-level 4 x 4 eliminations were 6 ms of host time per call of the shipped pipeline, r06).  a (V,4,4) ->
    (lu (V,4,4) float32, rows (V,4) int32)."""
    a = np.array(a, dtype=np.float32).reshape(-1, 4, 4).copy()
    V = a.shape[0]
    ar = np.arange(V)
    rows = np.tile(np.arange(4, dtype=np.int32), (V, 1))
    for k in range(4):
        p = k + np.argmax(np.abs(a[:, k:, k]), axis=1)
        if np.any(a[ar, p, k] == 0.0):
            raise ValueError("singular global2ego matrix")
        swap = p != k
        if swap.any():
            i_s, p_s = ar[swap], p[swap]
            tmp = a[i_s, k].copy(); a[i_s, k] = a[i_s, p_s]; a[i_s, p_s] = tmp
            rk = rows[i_s, k].copy(); rows[i_s, k] = rows[i_s, p_s]; rows[i_s, p_s] = rk
        for i in range(k + 1, 4):
            a[:, i, k] = a[:, i, k] / a[:, k, k]
            a[:, i, k + 1:] = a[:, i, k + 1:] - a[:, i, k][:, None] * a[:, k, k + 1:]
    return a, rows


def compose_choices(view_counts: Sequence[int], per_view: int, n_points: int, rng=np.random):
    """The two ``PointSample`` stages of the reference pipeline (points.py:411: ``np.random.choice(np.arange(len(points)),
    num_samples, replace=len(points) < num_samples)``), drawn from ``rng`` in the reference's order, composed into one index per
    output point: position in the concatenation, over the views, of each view's depth != 0 pixels in row-major order.
    A view whose depth map is all zero contributes nothing and draws nothing (points.py:335-336)."""
    firsts, kept = [], []
    off = 0
    for cnt in view_counts:
        cnt = int(cnt)
        if cnt > 0:
            ch = rng.choice(np.arange(cnt), per_view, replace=cnt < per_view)
            kept.append(ch.astype(np.int64) + off)
        off += cnt
    if not kept:
        raise ValueError("every depth map of the scene is empty")
    cat = np.concatenate(kept)                                   # AggregateMultiViewPoints: views in order
    ch2 = rng.choice(np.arange(len(cat)), n_points, replace=len(cat) < n_points)
    return cat[ch2]


class MultiViewIngest:
    """``MultiViewIngest(n_points)(scenes)``: scenes = list of dicts with the reference pipeline's keys

        depth_img      (V,H,W) GPU tensor, float32 metres (LoadDepthFromFile's output) or the decoded uint16 image
        depth_shift    divisor of a uint16 image (results['depth_shift']); ignored for float32
        depth_cam2img  (3,3) / (3,4) / (4,4) intrinsic, one for the scene or (V,...) per view
        extrinsic      (V,4,4) float32 global2ego (results['depth2img']['extrinsic'])
        aug            optional dict(rot_mat_T (3,3), scale, trans (3)): GlobalRotScaleTrans's effect on the points
        choices        optional precomputed (N,) index (see ``compose_choices``); otherwise drawn from ``rng``
    """

    def __init__(self, n_points: int = 100000, per_view_points: Optional[int] = None, depth_map_size=None,
                 use_color: bool = False):
        if depth_map_size is not None or use_color:
            raise NotImplementedError("ConvertRGBDToPoints(depth_map_size=..., use_color=True) is not part of the shipped "
                                      "pipeline (CFG:112, 134) and is not implemented on the device")
        self.n_points = int(n_points)
        self.per_view_points = int(per_view_points) if per_view_points is not None else self.n_points // 10   # CFG:113, 135

    @staticmethod
    def _intrinsics(k, V: int) -> np.ndarray:
        k = np.asarray(k.cpu() if isinstance(k, torch.Tensor) else k)
        if k.ndim == 2:
            k = np.broadcast_to(k, (V,) + k.shape)
        if k.shape[0] != V or k.shape[1] > 4 or k.shape[2] > 4:
            raise ValueError(f"depth_cam2img must be (r,c) or ({V},r,c) with r, c <= 4, got {k.shape}")
        pad = np.tile(np.eye(4, dtype=np.float32), (V, 1, 1))   # points_img2cam: pad to 4x4, invert (fp32), all views in one call
        pad[:, :k.shape[1], :k.shape[2]] = k.astype(np.float32)
        return np.linalg.inv(pad)

    # per-scene-slot scratch, kept across calls (a fresh pinned buffer per scene and call is a hipHostMalloc with an
    # implicit device synchronise each): the index workspace, the pinned per-view counts the host polls, and one pinned
    # staging buffer + its device twin for everything the gather needs from the host (composed choices, inverse intrinsics,
    # LU factors, pivots, augmentation) -- ONE asynchronous copy per scene
    class _Slot:
        __slots__ = ("key", "ws", "counts", "counts_np", "stage", "stage_np", "stage_dev", "copied")

    _MAX_SLOT_SETS = 4      # (thread, stream) pairs served concurrently; the least recently used set is retired (module._MAX_LANES)

    def _slot(self, b: int, V: int, H: int, W: int, N: int, dev, stream: int) -> "MultiViewIngest._Slot":
        # one set of slots per (thread, stream): two calls on different streams or from different threads (prefetch threads, serving
        # lanes) must not share a workspace, a staging buffer or the -1 preset of the pinned counts.  The table is an
        # LRU of at most _MAX_SLOT_SETS sets guarded by a lock (short-lived prefetch threads would otherwise pile up
        # pinned + device memory for the life of the object); a retired set's stream is drained first, its buffers go back to torch.
        with self.__dict__.setdefault("_slots_lock", threading.Lock()):
            per = self.__dict__.setdefault("_slots", {})
            skey = (threading.get_ident(), stream)
            slots = per.pop(skey, None)
            if slots is None:
                slots = []
                while len(per) >= self._MAX_SLOT_SETS:
                    old_key = next(iter(per))
                    for sl in per.pop(old_key):
                        if sl is not None and sl.copied is not None:
                            sl.copied.synchronize()
                    torch.cuda.synchronize(dev)      # kernels of the retired set may still read its workspace / staging twin
            per[skey] = slots                        # most recently used last
        while len(slots) <= b:
            slots.append(None)
        key = (V, H, W, N, str(dev))
        sl = slots[b]
        if sl is None or sl.key != key:
            lib = _abi.lib()
            sl = slots[b] = MultiViewIngest._Slot()
            sl.key = key
            sl.ws = torch.empty((lib.ptx_ingest_workspace_bytes(V, H, W),), dtype=torch.uint8, device=dev)
            sl.counts = torch.empty((max(V, 1),), dtype=torch.int32).pin_memory()
            sl.counts_np = sl.counts.numpy()
            nstage = (8 * N + 15) // 16 * 16 + 4 * (32 * V + 16) + 16 * V
            sl.stage = torch.empty((nstage,), dtype=torch.uint8).pin_memory()
            sl.stage_np = sl.stage.numpy()
            sl.stage_dev = torch.empty((nstage,), dtype=torch.uint8, device=dev)
            sl.copied = None
        return sl

    @torch.no_grad()
    def __call__(self, scenes: Sequence[Dict], rng=np.random) -> IngestedBatch:
        lib = _abi.lib()
        B, N = len(scenes), self.n_points
        if B == 0:
            raise ValueError("no scenes")
        dev = scenes[0]["depth_img"].device
        if dev.type != "cuda":
            raise RuntimeError("MultiViewIngest (HIP) needs GPU tensors: there is no CPU path")
        st = torch.cuda.current_stream(dev)
        out = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        bbox = torch.empty((B, 6), dtype=torch.int32, device=dev)
        # (cleared and set by ptx_ingest_gather: bit 0 = a rank beyond the scene's pixels.  The host has already checked the ranks
        #  against the published counts below, so the word is not read back -- the C ABI wants the buffer)
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        work = []
        for b, sc in enumerate(scenes):                          # 1. index every scene's depth maps (one pass each)
            depth = sc["depth_img"]
            if depth.device != dev or depth.dim() != 3 or depth.dtype not in _DEPTH_DTYPES or not depth.is_contiguous():
                raise RuntimeError(f"depth_img must be a contiguous (V,H,W) float32 / uint16 tensor on {dev}, got "
                                   f"{tuple(depth.shape)} {depth.dtype} on {depth.device}")
            V, H, W = depth.shape
            sl = self._slot(b, V, H, W, N, dev, st.cuda_stream)
            sl.counts_np[:V] = -1                                # published by the scan kernel with system scope
            _abi.check(lib.ptx_ingest_index(depth.data_ptr(), _DEPTH_DTYPES[depth.dtype], V, H, W, sl.ws.data_ptr(), sl.ws.numel(),
                                            sl.counts.data_ptr(), st.cuda_stream), "ptx_ingest_index")
            work.append((depth, sl))
        # the per-view matrices of ALL scenes in two batched numpy calls (inverse intrinsics, LU factors of global2ego): per scene they
        # were ~0.2 ms of Python between the launches of a chained call (r06: the ingest was host-bound at 0.3 ms per scene)
        Vs = [int(d.shape[0]) for d, _ in work]
        ks = [np.asarray(sc["depth_cam2img"].cpu() if isinstance(sc["depth_cam2img"], torch.Tensor) else sc["depth_cam2img"]) for sc in scenes]
        if len(set(Vs)) == 1 and all(k.shape == ks[0].shape for k in ks):
            stack = np.concatenate([np.broadcast_to(k, (Vs[0],) + k.shape) if k.ndim == 2 else k for k in ks])
            inv_all = self._intrinsics(stack, Vs[0] * B)
        else:
            inv_all = np.concatenate([self._intrinsics(k, V) for k, V in zip(ks, Vs)])
        ext_all = np.concatenate([np.asarray(sc["extrinsic"], dtype=np.float32).reshape(V, 4, 4) for sc, V in zip(scenes, Vs)])
        lus_all, pivs_all = lu_factor_4x4_batch(ext_all)
        voff = np.concatenate([[0], np.cumsum(Vs)])
        sels, vcs = [], []
        for b, (sc, (depth, sl)) in enumerate(zip(scenes, work)):           # 2. host RNG, 3. gather
            V, H, W = depth.shape
            # the per-view counts decide `replace` on the host: wait for THESE V words (not for the stream to drain)
            if lib.ptx_wait_counts(sl.counts.data_ptr(), V, 20_000_000) != 0:
                st.synchronize()
            vc = sl.counts_np[:V].copy()
            if vc.min() < 0:
                raise RuntimeError("ptx_ingest_index finished without publishing the per-view counts")
            sel = sc.get("choices")
            if sel is None:
                sel = compose_choices(vc, self.per_view_points, N, rng)
            sel = np.ascontiguousarray(sel, dtype=np.int64)
            if sel.shape != (N,):
                raise ValueError(f"choices must be ({N},), got {sel.shape}")
            if sel.min() < 0 or sel.max() >= int(vc.sum()):      # checked where the counts are: no device round trip for it
                raise IndexError(f"choices beyond the scene's depth != 0 pixels in scene {b}")
            inv_k, lus, pivs = inv_all[voff[b]:voff[b + 1]], lus_all[voff[b]:voff[b + 1]], pivs_all[voff[b]:voff[b + 1]]
            aug = sc.get("aug")
            if sl.copied is not None and not sl.copied.query():
                sl.copied.synchronize()                          # the previous call's copy out of the staging buffer (long done)
            o_sel, o_small = 0, (8 * N + 15) // 16 * 16          # 16-byte aligned tables
            o_piv = o_small + 4 * (32 * V + 16)
            sl.stage_np[o_sel:o_sel + 8 * N].view(np.int64)[:] = sel
            small = sl.stage_np[o_small:o_small + 4 * (32 * V + 13)].view(np.float32)
            small[:16 * V] = inv_k.reshape(-1)
            small[16 * V:32 * V] = lus.reshape(-1)
            if aug is not None:
                small[32 * V:32 * V + 9] = np.asarray(aug["rot_mat_T"], np.float32).reshape(9)
                small[32 * V + 9] = np.float32(aug["scale"])
                small[32 * V + 10:32 * V + 13] = np.asarray(aug["trans"], np.float32).reshape(3)
            sl.stage_np[o_piv:o_piv + 16 * V].view(np.int32)[:] = pivs.astype(np.int32).reshape(-1)
            sl.stage_dev.copy_(sl.stage, non_blocking=True)
            sl.copied = torch.cuda.Event()
            sl.copied.record(st)
            shift = float(sc.get("depth_shift", 1.0))
            dp = sl.stage_dev.data_ptr()
            fp = dp + o_small
            _abi.check(lib.ptx_ingest_gather(
                depth.data_ptr(), _DEPTH_DTYPES[depth.dtype], shift, V, H, W, fp, fp + 4 * V * 16, dp + o_piv,
                dp + o_sel, N, (fp + 4 * V * 32) if aug is not None else None, out[b].data_ptr(), bbox[b].data_ptr(),
                status[b:].data_ptr(), sl.ws.data_ptr(), sl.ws.numel(), st.cuda_stream), "ptx_ingest_gather")
            sels.append(sel)
            vcs.append(vc)
        return IngestedBatch(points=[out[b] for b in range(B)], bbox=bbox, view_counts=vcs, sel=sels)
