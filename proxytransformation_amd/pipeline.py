"""The detector's feature-extraction prefix around the neck, chained on ONE stream (BASELINE ``configs[3]``: "EmbodiedScan
mv-grounding config, full pipeline on 1xMI355X"; SURVEY 8f N4 -> the path -> N2 -> N3).

What ``SparseFeatureFusion3DGrounderPreshape.extract_feat`` does with a batch, minus the two learned backbones (2D ResNet and the
MinkowskiEngine ResNet are out of scope): reference lines on the left, the stage that runs here on the right.

    test_pipeline / train_pipeline  CFG:105-142              MultiViewIngest        depth maps -> (N,3) cloud + its bounding box
    self.preshape(points, text_dict, img_features[-1])  DET:385   module.forward(..., bbox=)   the hot path
    ME.utils.batch_sparse_collate + ME.SparseTensor  DET:388-397  module.quantize        (scene, floor(p / 0.01)) rows, first point kept
    x = self.backbone_3d(x); x[l].decomposed_coordinates[idx]  DET:398, 429-430   level_coordinates()   COORDINATES only, see below
    batch_point_sample(img_meta, img_features[l][idx], point, proj_mat, ...)  DET:431-444   fusion.batch_point_sample   per level

``backbone_3d`` (MinkResNet, backbones/mink_resnet.py:57-78: conv1 stride 2, max-pool stride 2, four stages of stride 2) is a learned
sparse CNN and is NOT rebuilt.  The only thing the point sampling needs from it are the COORDINATES of its four output levels, and
those do not depend on its weights: a strided MinkowskiEngine layer maps a coordinate c to floor(c / s) * s for its output tensor
stride s -- 8, 16, 32, 64 voxels for the four stages -- and keeps one row per distinct result.  ``level_coordinates`` produces exactly
that set (rows in first-occurrence order; ME's own row order is unspecified) by running the quantisation kernels on the integer voxel
rows (``ptx_voxel_coarsen``); the features that travel with them in the reference are the backbone's and stay out.

Nothing in ``__call__`` synchronises the device: the host waits only for small integers the kernels publish through pinned memory
(per-view pixel counts, survivor counts, voxel row counts) -- the list LENGTHS the reference obtains with blocking ``.item()`` /
``nonzero`` / ``unique`` calls -- and every stage is enqueued on the caller's current stream; only the channels-last copies of the
feature maps that the sampling gathers from (they do not depend on the points) run on a side stream beside the ingest, forked and
joined by events (``overlap_feature_layout=False``: on the caller's stream as well).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _abi
from .fusion import batch_point_sample, prepare_features_many, reverse_3d_flow
from .ingest import IngestedBatch, MultiViewIngest

__all__ = ["GroundingFeaturePrefix", "PrefixOutput", "level_coordinates", "projection_matrices", "MINK_RESNET_STRIDES"]

#: output tensor strides of MinkResNet's four stages (backbones/mink_resnet.py:57-78 with pool=True): 2 * 2 * 2^(i+1)
MINK_RESNET_STRIDES = (8, 16, 32, 64)


@dataclass
class PrefixOutput:
    ingested: IngestedBatch                     # N4: the clouds the neck consumed (+ bbox, per-view counts, composed choices)
    points: List[torch.Tensor]                  # the neck's output, DET:385: B x (N_b', 3)
    coordinates: torch.Tensor                   # DET:388-397: (Nv, 4) int32 rows (scene, ix, iy, iz)
    features: torch.Tensor                      # (Nv, 3) fp32: use_xyz_feat=True (CFG:43) -- the surviving point of each voxel
    scene_rows: List[int]                       # rows of scene b = [scene_rows[b-1], scene_rows[b])
    level_coords: List[List[torch.Tensor]] = field(default_factory=list)      # [level][scene] (n, 3) int32, tensor stride units
    level_points: List[List[torch.Tensor]] = field(default_factory=list)      # [level][scene] (n, 3) fp32 = coords * voxel_size
    points_imgfeats: List[List[torch.Tensor]] = field(default_factory=list)   # [scene][level] (n, C_l) fp32 (DET:445-448)
    stage_ms: Optional[Dict[str, float]] = None


def projection_matrices(depth2img: dict) -> np.ndarray:
    """``intrinsic @ extrinsic`` per view (DET:419-423) in fp32 on the host: (V,4,4)."""
    ext = [np.asarray(e, np.float32).reshape(4, 4) for e in depth2img["extrinsic"]]
    intr = depth2img["intrinsic"]
    if not isinstance(intr, (list, tuple)):                      # one matrix for every view (multiview.py:167-170)
        intr = [intr] * len(ext)
    out = np.empty((len(ext), 4, 4), np.float32)
    for v, (k, e) in enumerate(zip(intr, ext)):
        k4 = np.eye(4, dtype=np.float32)
        k = np.asarray(k, np.float32)
        k4[:k.shape[0], :k.shape[1]] = k
        out[v] = k4 @ e
    return out


class _CoarsenScratch:
    """Workspace + pinned count words of ``level_coordinates`` (one per pipeline object and stream, reused across calls)."""

    def __init__(self, B: int, ncap: int, dev):
        lib = _abi.lib()
        nbytes = lib.ptx_voxel_workspace_bytes(B, ncap)
        if nbytes == 0:
            raise RuntimeError(f"level_coordinates: unsupported size B={B}, rows per scene={ncap}")
        self.key = (B, ncap, str(dev))
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.info = torch.empty((2 + B,), dtype=torch.int32).pin_memory()
        self.info_np = self.info.numpy()


def level_coordinates(coordinates: torch.Tensor, scene_rows: Sequence[int], stride: int, voxel_size: float, scratch: dict = None):
    """Coordinates of a MinkowskiEngine level of tensor stride ``stride`` over the voxel rows ``coordinates`` (Nv,4) of a finer level
    (``scene_rows[b]`` = end of scene b's rows): per scene the distinct ``floor(c / stride) * stride`` in first-occurrence order (module
    docstring), and their positions ``coordinate * voxel_size`` (DET:429-430) -- one call of ``ptx_voxel_coarsen`` (csrc/voxel.hip: the
    quantisation kernels on integer rows).  Returns ``(coords (n,4) int32, points (n,3) fp32, ends)``; the host waits only for the row
    counts, which the kernel publishes through pinned memory."""
    import ctypes
    lib = _abi.lib()
    B = len(scene_rows)
    dev = coordinates.device
    lo = [0] + list(scene_rows[:-1])
    ncap = max(max(e - l for e, l in zip(scene_rows, lo)), 1)
    cap = 1 << (ncap - 1).bit_length()                             # workspace sized for the next power of two: reused across levels / calls
    scratch = {} if scratch is None else scratch
    st = torch.cuda.current_stream(dev)
    sc = scratch.get(st.cuda_stream)
    if sc is None or sc.key[0] != B or sc.key[1] < cap or sc.key[2] != str(dev):
        sc = scratch[st.cuda_stream] = _CoarsenScratch(B, cap, dev)
    total = int(scene_rows[-1])
    out_c = torch.empty((total, 4), dtype=torch.int32, device=dev)
    out_p = torch.empty((total, 3), dtype=torch.float32, device=dev)
    ends_in = (ctypes.c_int32 * B)(*[int(e) for e in scene_rows])
    sc.info_np[:] = -1
    _abi.check(lib.ptx_voxel_coarsen(coordinates.data_ptr(), ends_in, B, int(stride), float(voxel_size), out_c.data_ptr(), out_p.data_ptr(),
                                     sc.info.data_ptr(), sc.info.data_ptr() + 8, sc.ws.data_ptr(), sc.ws.numel(), st.cuda_stream),
               "ptx_voxel_coarsen")
    if lib.ptx_wait_counts(sc.info.data_ptr(), 2 + B, 20_000_000) != 0:
        st.synchronize()
    n, overflow = int(sc.info_np[0]), int(sc.info_np[1])
    if n < 0 or n == 0x7fffffff or overflow:
        raise RuntimeError(f"ptx_voxel_coarsen failed (rows {n}, overflow {overflow})")
    ends = sc.info_np[2:2 + B].tolist()
    return out_c[:n], out_p[:n], ends


class GroundingFeaturePrefix:
    """ingest -> preshape -> quantise -> level coordinates -> image-feature sampling, as one call (module docstring).

    ``scenes``: one dict per sample with what the loading transforms hand on -- ``depth_img`` (V,H,W) float32 metres or uint16 raw
    (+ ``depth_shift``), ``depth_cam2img``, ``depth2img = dict(extrinsic=[V x (4,4) global2cam], intrinsic=[V x cam2img] | one)``
    (mv_3dvg_dataset.py:544-553), optional ``aug`` / ``choices`` (ingest.MultiViewIngest) and ``img_meta`` (``scale_factor``, ``flip``,
    ``img_crop_offset``, ``img_shape``, the 3D flow record) for the sampling.  ``img_features``: the 2D backbone's levels, each
    (B,V,C_l,H_l,W_l); the LAST one feeds the neck (DET:385)."""

    def __init__(self, preshape, n_points: int = 100000, voxel_size: float = 0.01,
                 level_strides: Sequence[int] = MINK_RESNET_STRIDES, coord_type: str = "DEPTH", overlap_feature_layout: bool = True):
        self.preshape = preshape
        self.ingest = MultiViewIngest(n_points)
        self.voxel_size = float(voxel_size)
        self.level_strides = tuple(int(s) for s in level_strides)
        self.coord_type = coord_type
        self._stage = {}                # per stream: pinned staging of the per-scene projection matrices / reverse 3D flows + device twin
        self._coarsen = {}              # per stream: workspace + pinned count words of level_coordinates
        #: the channels-last copies of the feature maps that the sampling gathers from (~1 ms of HBM traffic at the shipped shape) are
        #: made on a side stream at the START of the call, beside the ingest -- whose per-scene host work leaves the GPU idle --
        #: and joined in front of the sampling (events, no host wait); False: inside each sampling call, on the caller's stream
        self.overlap_feature_layout = overlap_feature_layout
        self._side = None

    def _upload_matrices(self, scenes, dev, st):
        """Projection matrices (V,4,4) and the reverse 3D augmentation flow (3,4) of every scene through ONE pinned staging buffer and
        ONE asynchronous copy: a ``.to(device)`` from pageable memory would block the host behind everything queued on the stream."""
        projs = [projection_matrices(sc["depth2img"]) for sc in scenes]
        flows = [reverse_3d_flow(sc.get("img_meta") or {}, self.coord_type) for sc in scenes]
        n = sum(p.size for p in projs) + 12 * len(scenes)
        stg = self._stage.get(st.cuda_stream)           # per stream: the device twin is overwritten in stream order
        if stg is None or stg["host"].numel() < n or stg["dev"].device != dev:
            stg = self._stage[st.cuda_stream] = dict(host=torch.empty((n,), dtype=torch.float32).pin_memory(),
                                                     dev=torch.empty((n,), dtype=torch.float32, device=dev), done=None)
            stg["np"] = stg["host"].numpy()
        if stg["done"] is not None and not stg["done"].query():
            stg["done"].synchronize()                  # the previous call's copy out of the staging buffer (long done)
        o, proj_t, flow_t = 0, [], []
        for p, f in zip(projs, flows):
            stg["np"][o:o + p.size] = p.reshape(-1)
            proj_t.append(stg["dev"][o:o + p.size].view(p.shape))
            o += p.size
            if f is not None:
                stg["np"][o:o + 12] = f.numpy().reshape(-1)
                flow_t.append(stg["dev"][o:o + 12].view(3, 4))
            else:
                flow_t.append(None)
            o += 12
        stg["dev"][:o].copy_(stg["host"][:o], non_blocking=True)
        stg["done"] = torch.cuda.Event()
        stg["done"].record(st)
        return proj_t, flow_t

    @torch.no_grad()
    def __call__(self, scenes: Sequence[dict], text_dict: dict, img_features: Sequence[torch.Tensor],
                 img_pad_shape: Sequence[int] = (480, 480), rng=np.random, time_stages: bool = False) -> PrefixOutput:
        if len(img_features) != len(self.level_strides):
            raise ValueError(f"{len(self.level_strides)} levels of image features expected, got {len(img_features)}")
        dev = img_features[-1].device
        st = torch.cuda.current_stream(dev)
        marks = []

        def mark(name):
            if time_stages:
                e = torch.cuda.Event(enable_timing=True)
                e.record(st)
                marks.append((name, e))
        mark("start")
        prepared = None
        if self.overlap_feature_layout:
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(device=dev)
            self._side.wait_stream(st)                       # the feature maps were produced on the caller's stream
            nl = len(self.level_strides)
            with torch.cuda.stream(self._side):
                flat = prepare_features_many([img_features[li][b] for b in range(len(scenes)) for li in range(nl)])
            flat[0].record_stream(st)                        # ONE allocation under the side stream, consumed on the caller's
            prepared = [flat[b * nl:(b + 1) * nl] for b in range(len(scenes))]
        proj_t, flow_t = self._upload_matrices(scenes, dev, st)
        batch = self.ingest(scenes, rng=rng)                                               # N4
        mark("ingest")
        outs = self.preshape(batch.points, text_dict, img_features[-1], bbox=batch.bbox)   # the path, DET:385
        mark("preshape")
        coords, feats, ends = self.preshape.quantize(outs, self.voxel_size, return_scene_rows=True)   # N2, DET:388-397
        mark("quantize")
        res = PrefixOutput(ingested=batch, points=outs, coordinates=coords, features=feats, scene_rows=ends)
        cur_c, cur_e = coords, ends
        for s in self.level_strides:                                                       # DET:398, 429-430 (coordinates only)
            # every level from the one below it: floor(floor(c / 8) * 8 / 16) = floor(c / 16), on ever fewer rows
            cur_c, pts_l, cur_e = level_coordinates(cur_c, cur_e, s, self.voxel_size, self._coarsen)
            lo = [0] + cur_e[:-1]
            res.level_coords.append([cur_c[lo[b]:cur_e[b], 1:] for b in range(len(cur_e))])
            res.level_points.append([pts_l[lo[b]:cur_e[b]] for b in range(len(cur_e))])
        mark("levels")
        B = len(scenes)
        if prepared is not None:
            st.wait_stream(self._side)
        for b, sc in enumerate(scenes):                                                    # N3, DET:402-448
            meta = sc.get("img_meta") or {}
            proj = proj_t[b]
            per_level = []
            for li in range(len(self.level_strides)):
                per_level.append(batch_point_sample(
                    meta, img_features[li][b], res.level_points[li][b], proj, self.coord_type,
                    img_scale_factor=meta.get("scale_factor", (1.0, 1.0))[:2] if "scale_factor" in meta else 1.0,
                    img_crop_offset=meta.get("img_crop_offset", 0.0), img_flip=bool(meta.get("flip", False)),
                    img_pad_shape=tuple(img_pad_shape), img_shape=tuple(meta.get("img_shape", img_pad_shape))[:2],
                    aligned=False, pre_transform=flow_t[b], prepared=None if prepared is None else prepared[b][li]))
            res.points_imgfeats.append(per_level)
        mark("point_sample")
        assert len(res.points_imgfeats) == B
        if time_stages:
            marks[-1][1].synchronize()                 # the caller asked for times: this is the one place that waits
            res.stage_ms = {n: marks[i - 1][1].elapsed_time(e) for i, (n, e) in enumerate(marks) if i > 0}
            res.stage_ms["total"] = marks[0][1].elapsed_time(marks[-1][1])
        return res
