"""Image feature -> point sampling on the GPU: the reference's ``batch_point_sample``
(embodiedscan/models/layers/fusion_layers/point_fusion.py:208-313) with the argument list its detector uses
(detectors/sparse_featfusion_grounder_preshape.py:428-444), executed by ``ptx_point_sample`` (csrc/pointsample.hip)."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _abi

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _pair(v, dev):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().tolist()
    if isinstance(v, (int, float)):
        return float(v), float(v)
    return float(v[0]), float(v[1])


def reverse_3d_flow(img_meta: Optional[dict], coord_type: str = "DEPTH") -> Optional[torch.Tensor]:
    """``apply_3d_transformation(points, coord_type, img_meta, reverse=True)`` (point_fusion.py:20-107) as ONE (3,4) affine
    ``[A | t]`` (p' = A p + t), composed on the host in float64: the recorded ``transformation_3d_flow`` is undone back to
    front -- 'T': p - pcd_trans, 'S': p / pcd_scale_factor, 'R': p @ inverse(pcd_rotation), 'HF' / 'VF': the BEV flips of
    the coordinate type (DEPTH: x -> -x / y -> -y, depth_points.py:47-50; LIDAR: y -> -y / x -> -x, lidar_points.py:47-50;
    CAMERA: x -> -x / z -> -z, cam_points.py:47-50).  Returns
    ``None`` when nothing was recorded."""
    flow = list(img_meta.get("transformation_3d_flow", [])) if img_meta else []
    if not flow:
        return None
    import numpy as np
    A, t = np.eye(3), np.zeros(3)

    def host(x):
        return np.asarray(x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x, np.float64)
    rot = host(img_meta["pcd_rotation"]).reshape(3, 3) if "pcd_rotation" in img_meta else np.eye(3)
    scale = float(img_meta.get("pcd_scale_factor", 1.0))
    trans = host(img_meta["pcd_trans"]).reshape(3) if "pcd_trans" in img_meta else np.zeros(3)
    ct = coord_type.upper()
    if ct not in ("DEPTH", "LIDAR", "CAMERA"):
        raise ValueError(f"coord_type {coord_type!r}")
    for op in flow[::-1]:
        if op == "T":
            t = t - trans
        elif op == "S":
            A, t = A / scale, t / scale
        elif op == "R":                                  # row vectors: p @ inv(rot)  <=>  column vectors: inv(rot)^T p
            M = np.linalg.inv(rot).T
            A, t = M @ A, M @ t
        elif op in ("HF", "VF"):
            flipped = bool(img_meta.get("pcd_horizontal_flip" if op == "HF" else "pcd_vertical_flip", False))
            if flipped:
                # the axis each Points class negates (structures/points/depth_points.py:47-50, lidar_points.py:47-50,
                # cam_points.py:47-50): DEPTH x / y, LIDAR y / x, CAMERA x / z
                axis = {"DEPTH": (0, 1), "LIDAR": (1, 0), "CAMERA": (0, 2)}[ct][op == "VF"]
                F = np.eye(3)
                F[axis, axis] = -1.0
                A, t = F @ A, F @ t
        else:
            raise AssertionError(f"This 3D data transformation op ({op}) is not supported")     # point_fusion.py:101-102
    return torch.from_numpy(np.concatenate([A, t[:, None]], 1).astype(np.float32))


def batch_point_sample(img_meta: Optional[dict], img_features: torch.Tensor, points: torch.Tensor, proj_mat: torch.Tensor,
                       coord_type: str = "DEPTH", img_scale_factor=1.0, img_crop_offset=0.0, img_flip: bool = False,
                       img_pad_shape: Sequence[int] = (480, 640), img_shape: Sequence[int] = (480, 640),
                       aligned: bool = False, padding_mode: str = "zeros", align_corners: bool = True,
                       valid_flag: bool = True, pre_transform: Optional[torch.Tensor] = None,
                       prepared: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Same arguments as the reference function.  img_features (V,C,H,W) on the GPU (fp32 / bf16 / fp16), points (N,3)
    fp32, proj_mat (V,4,4) = intrinsic @ extrinsic.  Returns (N,C) fp32.

    ``aligned=False`` (the detector's call: nearest) and ``aligned=True`` (bilinear) are both HIP; zeros padding,
    align_corners and valid_flag as the detector passes them.  The reverse 3D augmentation of
    ``apply_3d_transformation`` (point_fusion.py:20-107) is composed from ``img_meta['transformation_3d_flow']``
    (``reverse_3d_flow``; the training pipeline's GlobalRotScaleTrans records 'R', 'S', 'T') unless an explicit
    ``pre_transform`` (3,4) is given.  ``prepared``: the workspace ``prepare_features(img_features)`` returned (the channels-last copy
    made earlier, e.g. on another stream): the call then only samples."""
    if padding_mode != "zeros" or not align_corners or not valid_flag:
        raise NotImplementedError("HIP path: padding_mode='zeros', align_corners=True, valid_flag=True "
                                  "(the call at sparse_featfusion_grounder_preshape.py:428-444)")
    if pre_transform is None:
        pre_transform = reverse_3d_flow(img_meta, coord_type)
    if not (img_features.is_cuda and points.is_cuda and proj_mat.is_cuda):
        raise RuntimeError("batch_point_sample (HIP) needs GPU tensors: there is no CPU path")
    if img_features.dtype not in _DT:
        img_features = img_features.float()
    V, C, H, W = img_features.shape
    feats = img_features.contiguous()
    pts = points.detach().to(torch.float32).contiguous()
    proj = proj_mat.detach().to(torch.float32).contiguous()
    if proj.shape != (V, 4, 4) or pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError(f"expected points (N,3) and proj_mat ({V},4,4), got {tuple(pts.shape)}, {tuple(proj.shape)}")
    N = pts.shape[0]
    dev = pts.device
    out = torch.empty((N, C), dtype=torch.float32, device=dev)
    if N == 0:
        return out
    lib = _abi.lib()
    nbytes = lib.ptx_point_sample_workspace_bytes(V, C, H, W)
    if nbytes == 0:
        raise RuntimeError(f"unsupported feature shape {tuple(img_features.shape)} (C <= 512)")
    if prepared is not None:
        if prepared.numel() < nbytes or prepared.device != dev:
            raise RuntimeError("prepared: not the workspace of prepare_features() for these feature maps")
        ws, feats_ptr = prepared, None
    else:
        ws, feats_ptr = torch.empty((nbytes,), dtype=torch.uint8, device=dev), feats.data_ptr()
    sw, sh = _pair(img_scale_factor, dev)
    cw, ch = _pair(img_crop_offset, dev)
    pre = None if pre_transform is None else pre_transform.detach().to(device=dev, dtype=torch.float32).contiguous()
    _abi.check(lib.ptx_point_sample(pts.data_ptr(), N, feats_ptr, _DT[feats.dtype], V, C, H, W, proj.data_ptr(),
                                    None if pre is None else pre.data_ptr(), sw, sh, cw, ch, 1 if img_flip else 0,
                                    float(img_shape[1]), float(img_pad_shape[0]), float(img_pad_shape[1]), 1 if aligned else 0,
                                    out.data_ptr(),
                                    None, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream),
               "ptx_point_sample")
    return out


def prepare_features(img_features: torch.Tensor) -> torch.Tensor:
    """The channels-last copy of one sample's feature maps (V,C,H,W) that ``batch_point_sample`` gathers from, made on the CURRENT
    stream: returns the workspace to pass as ``prepared=``.  The detector knows the feature maps (2D backbone) long before it knows
    the points (neck + sparse backbone): ``pipeline.GroundingFeaturePrefix`` makes these copies on a side stream beside the ingest."""
    if not img_features.is_cuda:
        raise RuntimeError("prepare_features (HIP) needs GPU tensors: there is no CPU path")
    if img_features.dtype not in _DT:
        img_features = img_features.float()
    feats = img_features.contiguous()
    V, C, H, W = feats.shape
    lib = _abi.lib()
    nbytes = lib.ptx_point_sample_workspace_bytes(V, C, H, W)
    if nbytes == 0:
        raise RuntimeError(f"unsupported feature shape {tuple(feats.shape)} (C <= 512)")
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=feats.device)
    _abi.check(lib.ptx_point_sample_prepare(feats.data_ptr(), _DT[feats.dtype], V, C, H, W, ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream(feats.device).cuda_stream), "ptx_point_sample_prepare")
    return ws


def prepare_features_many(maps) -> list:
    """``prepare_features`` for a list of (V,C,H,W) feature maps with ONE allocation (the pipeline prepares scenes x levels maps at the
    start of a call: 24 allocations + calls were 0.3 ms of host time in front of the ingest).  Returns one workspace view per map."""
    lib = _abi.lib()
    maps = [m if m.dtype in _DT else m.float() for m in maps]
    maps = [m.contiguous() for m in maps]
    sizes = []
    for m in maps:
        if not m.is_cuda:
            raise RuntimeError("prepare_features (HIP) needs GPU tensors: there is no CPU path")
        V, C, H, W = m.shape
        n = lib.ptx_point_sample_workspace_bytes(V, C, H, W)
        if n == 0:
            raise RuntimeError(f"unsupported feature shape {tuple(m.shape)} (C <= 512)")
        sizes.append(n)                                   # multiples of 256 bytes
    dev = maps[0].device
    flat = torch.empty((sum(sizes),), dtype=torch.uint8, device=dev)
    views = list(flat.split_with_sizes(sizes))
    st = torch.cuda.current_stream(dev).cuda_stream
    for m, ws in zip(maps, views):
        V, C, H, W = m.shape
        _abi.check(lib.ptx_point_sample_prepare(m.data_ptr(), _DT[m.dtype], V, C, H, W, ws.data_ptr(), ws.numel(), st),
                   "ptx_point_sample_prepare")
    return views
