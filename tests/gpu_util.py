"""Thin test-side wrapper that calls the stage entry points of the C ABI with torch tensors."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from proxytransformation_amd import _abi


def dev():
    return torch.device("cuda:0")


def t(x, dtype=None):
    a = torch.from_numpy(np.ascontiguousarray(x)).to(dev())
    return a if dtype is None else a.to(dtype)


class Stages:
    """Stage-level access to libproxyt_hip.so for one module (weights, prep tables, workspace)."""

    def __init__(self, module, B, N, L, V):
        self.m = module
        self.lib = _abi.lib()
        self.shape = module._shape(B, N, L, V)
        self.stream = torch.cuda.current_stream().cuda_stream
        module._ensure_prepared(self.shape, dev(), self.stream)
        self.ws = module._workspace(module._lane(dev(), torch.cuda.current_stream()), self.shape, dev(), self.stream)
        self.w, self.prep, self.lin = module._wstruct, module._prep, module._lin
        self.M = module.num_cluster
        self.K = module.num_sub

    def _s(self):
        return ctypes.byref(self.shape)

    def grid_centers(self, points):
        B, N, _ = points.shape
        mm = torch.empty((B, 2, 3), device=dev())
        c = torch.empty((B, self.M, 3), device=dev())
        _abi.check(self.lib.ptx_grid_centers(points.data_ptr(), B, N, self.lin.data_ptr(), self.m.grid_size,
                                             4.0, mm.data_ptr(), c.data_ptr(), self.ws.data_ptr(),
                                             self.ws.numel(), self.stream), "ptx_grid_centers")
        return mm, c

    def ball_query(self, centers, points, K=None, radius=3.0):
        B, M, _ = centers.shape
        N = points.shape[1]
        K = K or self.K
        idx = torch.empty((B, M, K), dtype=torch.int32, device=dev())
        cl = torch.empty((B, M, K, 3), device=dev())
        pc = torch.empty((B, M), dtype=torch.int32, device=dev())
        _abi.check(self.lib.ptx_ball_query(centers.data_ptr(), points.data_ptr(), B, M, N, K, radius,
                                           idx.data_ptr(), cl.data_ptr(), pc.data_ptr(), self.stream),
                   "ptx_ball_query")
        return idx, cl, pc

    def offset_net(self, centers, cluster, minmax):
        out = torch.empty_like(centers)
        off = torch.empty_like(centers)
        _abi.check(self.lib.ptx_offset_net(self._s(), ctypes.byref(self.w), self.prep.data_ptr(),
                                           centers.data_ptr(), cluster.data_ptr(), minmax.data_ptr(),
                                           out.data_ptr(), off.data_ptr(), self.stream), "ptx_offset_net")
        return out, off

    def select(self, idx, centers, cluster, pad_count, order_override=None):
        s = self.shape
        B, Kd = s.B, s.Mt - s.Mk
        i32 = dict(dtype=torch.int32, device=dev())
        o = dict(order=torch.empty((B, s.Mt), **i32), picks=torch.empty((B, max(Kd, 1)), **i32),
                 keep=torch.empty((B, s.Mk), **i32), kcenter=torch.empty((B, s.Mk, 3), device=dev()),
                 kcluster=torch.empty((B, s.Mk, s.K, 3), device=dev()),
                 kidx=torch.empty((B, s.Mk, s.K), **i32), drop_idx=torch.empty((B, max(Kd, 1) * s.K), **i32),
                 tag=torch.full((B, s.N), 0x5a5a5a5a, **i32))      # garbage on entry: ptx_select_clusters writes every word (r05)
        _abi.check(self.lib.ptx_select_clusters(
            self._s(), idx.data_ptr(), centers.data_ptr(), cluster.data_ptr(), pad_count.data_ptr(),
            None if order_override is None else order_override.data_ptr(),
            o["order"].data_ptr(), o["picks"].data_ptr(), o["keep"].data_ptr(), o["kcenter"].data_ptr(),
            o["kcluster"].data_ptr(), o["kidx"].data_ptr(), o["drop_idx"].data_ptr(), o["tag"].data_ptr(),
            self.stream), "ptx_select_clusters")
        o["picks"] = o["picks"][:, :Kd]
        o["drop_idx"] = o["drop_idx"][:, :Kd * s.K]
        return o

    def pointnet(self, kcenter, kcluster):
        s = self.shape
        pp = torch.empty((s.B, s.Mk, s.C), device=dev())
        _abi.check(self.lib.ptx_pointnet(self._s(), ctypes.byref(self.w), self.prep.data_ptr(),
                                         kcenter.data_ptr(), kcluster.data_ptr(), pp.data_ptr(), self.stream),
                   "ptx_pointnet")
        return pp

    def img_proxy(self, img):
        s = self.shape
        out = torch.empty((s.B, s.V, s.C), device=dev())
        _abi.check(self.lib.ptx_img_proxy(self._s(), ctypes.byref(self.w), self.prep.data_ptr(), img.data_ptr(),
                                          out.data_ptr(), self.ws.data_ptr(), self.ws.numel(), self.stream),
                   "ptx_img_proxy")
        return out

    def proxy_block(self, which, point_proxy, proxy, mask=None):
        s = self.shape
        nout = 3 if which == 0 else 9
        head = torch.empty((s.B, s.Mk, nout), device=dev())
        guide = torch.empty((s.B, s.Mk, s.C), device=dev())
        _abi.check(self.lib.ptx_proxy_block(
            self._s(), ctypes.byref(self.w), self.prep.data_ptr(), which, point_proxy.data_ptr(),
            proxy.data_ptr(), proxy.shape[1], None if mask is None else mask.data_ptr(), head.data_ptr(),
            guide.data_ptr(), self.ws.data_ptr(), self.ws.numel(), self.stream), "ptx_proxy_block")
        return head, guide

    def affine_scatter(self, points, tag, kcenter, translate, transform):
        out = torch.empty_like(points)
        _abi.check(self.lib.ptx_affine_scatter(self._s(), points.data_ptr(), tag.data_ptr(), kcenter.data_ptr(),
                                               translate.data_ptr(), transform.data_ptr(), out.data_ptr(),
                                               self.stream), "ptx_affine_scatter")
        return out

    def affine_compact(self, points, tag, kcenter, translate, transform):
        out = torch.empty_like(points)
        counts = torch.empty((points.shape[0],), dtype=torch.int32, device=dev())
        _abi.check(self.lib.ptx_affine_compact(self._s(), points.data_ptr(), tag.data_ptr(), kcenter.data_ptr(),
                                               translate.data_ptr(), transform.data_ptr(), out.data_ptr(),
                                               counts.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
                                               self.stream), "ptx_affine_compact")
        n = counts.cpu().tolist()
        return [out[b, :n[b]] for b in range(points.shape[0])]
