"""``ProxyTransformationNormReverse`` for MI355X: the reference's registry entry,
constructor, ``state_dict`` layout and ``forward`` contract, executed by the
hand-written HIP kernels of ``libproxyt_hip.so`` through a ctypes C ABI.

Reference: embodiedscan/models/necks/preshape_norm_reverse_drop.py (PRE)
  * class + registration          PRE:280-281
  * constructor arguments         PRE:282-285 (kept verbatim, including the
                                  ``dynamic_drop_radio`` / ``mlp_radio`` / ``img_spacial_dim`` spellings)
  * parameter names / shapes      PRE:22-330 (checkpoints of the reference load unchanged)
  * forward(points, text_dict, img_feat) -> list of (N_i', 3) tensors    PRE:424-469
  * caller                        detectors/sparse_featfusion_grounder_preshape.py:95, 385

The torch modules declared below are *parameter containers only* -- none of their
``forward`` methods is ever called.  All arithmetic of the path runs in HIP; if the
extension is missing, or the inputs are not on a GPU, ``forward`` raises.
"""
from __future__ import annotations

import atexit
import ctypes
import os
import sys
import weakref
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _abi
from .registry import MODELS

__all__ = ["ProxyTransformationNormReverse"]

_RADIUS, _MARGIN = 3.0, 4.0          # PRE:23 (fixed, not reachable from the config)
_EMPTY_DROP = 0.3                    # PRE:352
_SLOT_WIDTH = 256                    # PRE:31, PRE:302 (hard-coded in the reference)
_EMBED_DIMS = (256, 512)             # 256 = everything the reference can run; 512 = BASELINE's stress config


def _bias_grid(dim: int) -> int:
    """Side of the per-slot positional-bias grid: int(sqrt(dim)) for a perfect square (PRE:196); a dim that is
    no perfect square -- which the reference cannot run at all (its reshape at PRE:217 fails) -- uses the
    next larger grid, of which the first ``dim`` entries are taken."""
    s = int(dim ** 0.5)
    return s if s * s == dim else s + 1

_IMG_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}   # PtxShape.img_dtype
_COMPUTE_DTYPES = {"fp32": 0, "bf16": 1}                                # PtxForwardOpts.compute_dtype
_COUNTS_TIMEOUT_US = 20_000_000         # then fall back to a stream synchronise
_MAX_SCENES_PER_CALL = 32            # kMaxScenes of the C ABI (per-scene pointer table passed by value)


# --------------------------------------------------------------------------- containers
class _Holder(nn.Module):
    """A module that only owns parameters / sub-modules (never called)."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("parameter container: the HIP path does not call torch sub-modules")


def _slot_mlp(hidden: int) -> nn.Sequential:
    # Conv2d(6,hidden,1) + BatchNorm2d + ReLU                         PRE:72-76 / PRE:112-116
    return nn.Sequential(nn.Conv2d(6, hidden, 1), nn.BatchNorm2d(hidden), nn.ReLU())


class _OffsetNetwork(_Holder):                                        # PRE:69-78
    def __init__(self, hidden: int):
        super().__init__()
        self.mlp = _slot_mlp(hidden)
        self.channel_mapper = nn.Conv1d(hidden, 3, kernel_size=1, bias=False)


class _DeformablePointCluster(_Holder):                               # PRE:22-31
    def __init__(self, hidden: int):
        super().__init__()
        self.get_offsets = _OffsetNetwork(hidden)


class _SimplifiedPointNet(_Holder):                                   # PRE:109-117
    def __init__(self, hidden: int):
        super().__init__()
        self.mlp = _slot_mlp(hidden)


class _AttentionPool2d(_Holder):                                      # PRE:144-152
    def __init__(self, spacial_dim: int, dim: int):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, dim) / dim ** 0.5)
        self.k_proj = nn.Linear(dim, dim)
        self.q_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.c_proj = nn.Linear(dim, dim)


class _ProxyAttention(_Holder):                                       # PRE:179-204
    def __init__(self, dim: int, kept: int, qkv_bias: bool):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proxy_proj = nn.Linear(dim, dim)
        self.proj = nn.Linear(dim, dim)
        s = _bias_grid(dim)
        self.pb_bias = nn.Parameter(torch.zeros(1, kept, 4, 4))
        self.pc_bias = nn.Parameter(torch.zeros(1, kept, s, 1))
        self.pr_bias = nn.Parameter(torch.zeros(1, kept, 1, s))
        for p in (self.pb_bias, self.pc_bias, self.pr_bias):
            nn.init.trunc_normal_(p, std=.02, a=-.04, b=.04)


class _Mlp(_Holder):                                                   # timm Mlp: fc1, fc2
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _ProxyBlock(_Holder):                                            # PRE:259-271
    def __init__(self, dim: int, hidden: int, kept: int, qkv_bias: bool):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _ProxyAttention(dim, kept, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, hidden)


class _Lane:
    """Per-(device, stream) call state: the library context (side streams / events), one workspace and the pinned
    count buffer.  Calls on different torch streams use different lanes, so independent batches can be in flight
    on the GPU at the same time (a serving loop alternating between two streams overlaps the bandwidth-bound
    image passes of one batch with the latency-bound proxy blocks of the other)."""
    __slots__ = ("ctx", "ws", "ws_key", "ws_dirty", "counts", "counts_np", "stream", "quant", "unchecked")

    def __init__(self, stream):
        self.stream = stream
        self.ctx = ctypes.c_void_p()
        _abi.check(_abi.lib().ptx_context_create(ctypes.byref(self.ctx)), "ptx_context_create")
        self.ws, self.ws_key, self.ws_dirty = None, None, True
        self.counts = self.counts_np = None
        self.quant = None                   # scratch of module.quantize (voxel hash, pinned count buffers)
        self.unchecked = False              # a forward was handed out whose join gate has not been checked behind a drain yet

    def sync_check(self) -> Optional[str]:
        """Drain this lane's streams and ask the library whether a stream gate of a forward issued on it failed (a join gate fails
        AFTER the survivor counts are out, i.e. after forward() has returned): the error text, or None."""
        if self.ctx is None or not self.unchecked:
            return None
        self.unchecked = False
        if _abi.lib().ptx_context_sync_check(self.ctx) != 0:
            self.ws_dirty = True            # the clean-on-entry words of that forward cannot be trusted
            return _abi.lib().ptx_last_error().decode()
        return None

    def release(self):
        ctx, self.ctx = self.ctx, None
        if ctx is not None:
            _abi.lib().ptx_context_destroy(ctx)
        self.ws = None


_MAX_LANES = 4                       # streams served concurrently by one module (least recently used is retired)


# Every live module is known to the interpreter-exit hook below: a forward whose join gate failed after forward() returned (outputs
# NaN, csrc/api.hip "gates") must not let the process end quietly with exit code 0 when it was the LAST forward of a loop.
_LIVE_MODULES = weakref.WeakSet()
_ORPHAN_LANES: list = []


def _exit_check():
    """Interpreter-exit hook.  When a failure was never reported it ends the process with ``os._exit(70)``: a HARD exit -- an atexit
    handler cannot change the exit status any other way, and a silent 0 is the worse outcome.  atexit runs handlers last-in
    first-out and this one is registered at import of the package, so handlers registered later (process-group teardown, loggers)
    have already run; stdout / stderr are flushed here; anything else buffered at that point is lost (documented in INTEGRATION.md)."""
    failed = []
    while _ORPHAN_LANES:                # lanes of modules that were collected inside a stream capture (__del__)
        lane = _ORPHAN_LANES.pop()
        try:
            err = lane.sync_check()
            if err:
                failed.append(err)
            lane.release()
        except Exception:
            pass
    for mod in list(_LIVE_MODULES):
        try:
            mod.check()
        except RuntimeError as e:
            failed.append(str(e))
        except Exception:               # interpreter shutdown: the runtime may already be gone
            pass
    if failed:
        sys.stderr.write("proxytransformation_amd: UNREPORTED FAILURE at interpreter exit -- " + " | ".join(failed) + "\n")
        sys.stderr.flush()
        try:
            sys.stdout.flush()
        finally:
            os._exit(70)                # EX_SOFTWARE: results of this process are not to be trusted


atexit.register(_exit_check)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _invalidate_after_load(mod, _incompatible_keys):
    mod.invalidate_weights()


# --------------------------------------------------------------------------- the module
@MODELS.register_module()
class ProxyTransformationNormReverse(nn.Module):
    """Drop-in for the reference neck of the same name (see module docstring)."""
    _instances = 0

    def __init__(self, embed_dim=256, num_heads=8, n_points=100000, grid_size=4, text_blocks=1,
                 img_blocks=1, dynamic_drop_radio=0.8, mlp_radio=4, qkv_bias=False, drop_rate=0.2,
                 attn_drop_rate=0.2, drop_path_rate=0.2, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 num_sub=30, drop_radio=0.2, input_dim=512, img_spacial_dim=15, *, compute_dtype="fp32"):
        super().__init__()
        if compute_dtype not in _COMPUTE_DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(_COMPUTE_DTYPES)} (got {compute_dtype!r})")
        #: arithmetic of the ProxyBlock GEMMs / attention in EVAL mode (extra, keyword-only; not in the reference):
        #: "fp32" = fp32-equivalent (the parity path), "bf16" = plain bf16 operands with fp32 accumulation -- what the
        #: reference's linears run in under ``--amp`` (tools/train.py:93-105); outputs then differ by ~1e-2 m
        self.compute_dtype = compute_dtype
        if act_layer is not nn.GELU or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("the HIP path implements act_layer=nn.GELU, norm_layer=nn.LayerNorm")
        # r04: 4, 8 (the reference's default, PRE:282) or 16 heads of head_dim 32 / 64: (256, 4 | 8), (512, 8 | 16)
        if embed_dim not in _EMBED_DIMS or num_heads not in (4, 8, 16) or embed_dim // num_heads not in (32, 64):
            raise NotImplementedError(f"the HIP path implements embed_dim in {_EMBED_DIMS} with 4, 8 or 16 heads of head_dim 32 or 64 "
                                      f"(got embed_dim={embed_dim}, num_heads={num_heads})")
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.grid_size = grid_size
        self.num_cluster = grid_size ** 3
        self.num_sub = num_sub or n_points // self.num_cluster               # PRE:291
        self.input_dim = input_dim
        self.img_spacial_dim = img_spacial_dim
        self.drop_radio = drop_radio
        self.text_blocks = text_blocks
        self.img_blocks = img_blocks
        self.dynamic_drop_radio = dynamic_drop_radio
        self.mlp_hidden = int(embed_dim * mlp_radio)
        self.drop_rate, self.attn_drop_rate, self.drop_path_rate = drop_rate, attn_drop_rate, drop_path_rate
        kept = int(self.num_cluster * (1 - dynamic_drop_radio))              # PRE:195
        self.real_cluster_num = kept

        self.get_deformable_cluster = _DeformablePointCluster(_SLOT_WIDTH)
        # the reference hard-codes 256 output channels (PRE:302), which only works with embed_dim = 256;
        # the point encoder is as wide as the tokens it feeds
        self.simple_encoder = _SimplifiedPointNet(embed_dim)
        self.channel_mapper = nn.Conv2d(input_dim, embed_dim, kernel_size=1)
        self.attn_pool2d = _AttentionPool2d(img_spacial_dim, embed_dim)
        self.norm_img = nn.LayerNorm(embed_dim)
        self.textformer = nn.ModuleList(
            [_ProxyBlock(embed_dim, self.mlp_hidden, kept, qkv_bias) for _ in range(text_blocks)])
        self.text_norm = nn.ModuleList([nn.LayerNorm(embed_dim) for _ in range(text_blocks)])
        self.imgformer = nn.ModuleList(
            [_ProxyBlock(embed_dim, self.mlp_hidden, kept, qkv_bias) for _ in range(img_blocks)])
        self.img_norm = nn.ModuleList([nn.LayerNorm(embed_dim) for _ in range(img_blocks)])
        self.text_trans = nn.Linear(embed_dim, 3)
        self.img_trans = nn.Linear(embed_dim, 9)
        self.text_trans_norm = nn.BatchNorm1d(3)
        self.img_trans_norm = nn.BatchNorm1d(9)

        # host-side caches (not part of the state_dict)
        self._tensors = None
        self._slots = None
        self._lanes: Dict[tuple, _Lane] = {}
        self._train_calls = 0
        self._train_checked = None
        self._train_live = None              # train mode: the parameters that receive a gradient (one-node step)
        self._train_static = None            # train mode: sub-modules / parameter tuples of the one-node step (train._static)
        self._train_side = None              # train mode: side stream of the image branch, per device
        self._train_pin = None               # train mode: pinned words + event of the early count read-back, per batch size
        ProxyTransformationNormReverse._instances += 1
        self._instance_salt = ProxyTransformationNormReverse._instances      # dropout masks differ between instances
        _LIVE_MODULES.add(self)
        self._warned_eval_grad = False
        self._lin_t = None
        # stochastic-depth rate of the blocks that are live (the last of each list; PRE:298-299)
        self._text_dpr = float(torch.linspace(0, drop_path_rate, text_blocks)[-1])
        self._img_dpr = float(torch.linspace(0, drop_path_rate, img_blocks)[-1])
        self.register_load_state_dict_post_hook(_invalidate_after_load)
        #: True = block until the whole forward has drained (the pre-ABI-3 behaviour); default is
        #: to return once the output lengths are known, like any asynchronous torch op
        self.sync_outputs = False
        self._wkey = None
        self._wstruct: Optional[_abi.PtxWeights] = None
        self._prep: Optional[torch.Tensor] = None
        self._lin: Optional[torch.Tensor] = None
        self._shapes: Dict[tuple, _abi.PtxShape] = {}
        # test-only hooks: replay a captured argsort / inject clamped centres
        self._order_override: Optional[torch.Tensor] = None
        self._centers_override: Optional[torch.Tensor] = None

    # host caches that hold ctypes pointers / device scratch: never copied or pickled (copy.deepcopy(model),
    # torch.save(model), EMA / SWA copies made after the first forward); a copy rebuilds them on its first call
    _HOST_CACHES = dict(_tensors=None, _slots=None, _lanes=None, _lin_t=None, _wkey=None, _wstruct=None, _prep=None,
                        _lin=None, _shapes=None, _graph_keepalive=None, _train_pin=None, _train_side=None, _train_live=None,
                        _train_mods=None, _train_static=None)

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._HOST_CACHES:
            state[k] = {} if k in ("_lanes", "_shapes") else None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        ProxyTransformationNormReverse._instances += 1
        self._instance_salt = ProxyTransformationNormReverse._instances
        _LIVE_MODULES.add(self)

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__(copy.deepcopy(self.__getstate__(), memo))
        return new

    def _dpr_last(self, blk) -> float:
        return self._text_dpr if blk is self.textformer[-1] else self._img_dpr

    def _train_lin(self, device):
        if self._lin_t is None or self._lin_t.device != device:
            self._lin_t = torch.linspace(0, 1, self.grid_size, device="cpu").to(device)      # PRE:41
        return self._lin_t

    # ------------------------------------------------------------------ reference-named helpers
    def get_text_proxy(self, text_dict):                                       # PRE:332-333
        return text_dict.values()

    # ------------------------------------------------------------------ shape / weights plumbing
    def _shape(self, B: int, N: int, L: int, V: int, img_dtype: int = 0) -> _abi.PtxShape:
        M = self.num_cluster
        return _abi.PtxShape(B=B, N=N, grid_size=self.grid_size, K=self.num_sub,
                             Mt=M - int(M * _EMPTY_DROP), Mk=self.real_cluster_num, L=L, V=V,
                             C=self.embed_dim, heads=self.num_heads, hidden=self.mlp_hidden,
                             in_dim=self.input_dim, hw=self.img_spacial_dim ** 2, img_dtype=img_dtype,
                             radius=_RADIUS, margin=_MARGIN, bn_eps=self.text_trans_norm.eps,
                             ln_eps=self.norm_img.eps)

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() re-allocate parameter storage: drop every cached pointer
        self.invalidate_weights()
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode: bool = True):
        # mode switches are where frameworks swap weights behind autograd's back (mmengine's EMAHook copies
        # through ``.data`` right before ``model.eval()``): re-derive the parameter tables afterwards
        self.invalidate_weights()
        return super().train(mode)

    def invalidate_weights(self) -> None:
        """Force the parameter-derived tables (folded BatchNorm, per-slot bias tables, folded attention-pool
        matrices) to be rebuilt on the next forward.  Needed only after writes that autograd cannot see
        (``p.data.copy_()``, ``p.data.normal_()``): every other change -- optimiser steps, ``load_state_dict``
        (also with ``assign=True``), re-assigned ``nn.Parameter`` objects, ``.to()``, ``train()`` / ``eval()`` --
        is detected automatically."""
        self._tensors = None
        self._slots = None
        self._wkey = None
        self._train_checked = None

    def _weights_key(self):
        """Per-call change detector over the LIVE parameter / buffer objects: the owning ``_parameters`` /
        ``_buffers`` dicts are read on every call, so re-assigned tensors (``load_state_dict(assign=True)``,
        ``mod.x.weight = nn.Parameter(...)``) are seen; in-place updates bump ``_version``; storage swaps
        change ``data_ptr``.  ~20 us of host time per call."""
        if self._slots is None:
            slots = []
            for mod in self.modules():
                slots += [(mod._parameters, k) for k, v in mod._parameters.items() if v is not None]
                slots += [(mod._buffers, k) for k, v in mod._buffers.items() if v is not None]
            self._slots = slots
        cur = [d[k] for d, k in self._slots]
        self._tensors = cur                       # keeps the ids below from being recycled
        return tuple([id(t) for t in cur]), tuple([t._version for t in cur]), tuple([t.data_ptr() for t in cur])

    def _block_struct(self, blk: _ProxyBlock, out_norm: nn.LayerNorm) -> _abi.PtxBlock:
        a = blk.attn
        return _abi.PtxBlock(
            norm1_w=_ptr(blk.norm1.weight), norm1_b=_ptr(blk.norm1.bias),
            pb_bias=_ptr(a.pb_bias), pc_bias=_ptr(a.pc_bias), pr_bias=_ptr(a.pr_bias),
            qkv_w=_ptr(a.qkv.weight), qkv_b=_ptr(a.qkv.bias),
            pp_w=_ptr(a.proxy_proj.weight), pp_b=_ptr(a.proxy_proj.bias),
            proj_w=_ptr(a.proj.weight), proj_b=_ptr(a.proj.bias),
            norm2_w=_ptr(blk.norm2.weight), norm2_b=_ptr(blk.norm2.bias),
            fc1_w=_ptr(blk.mlp.fc1.weight), fc1_b=_ptr(blk.mlp.fc1.bias),
            fc2_w=_ptr(blk.mlp.fc2.weight), fc2_b=_ptr(blk.mlp.fc2.bias),
            out_norm_w=_ptr(out_norm.weight), out_norm_b=_ptr(out_norm.bias))

    @staticmethod
    def _slot_struct(seq: nn.Sequential) -> _abi.PtxSlotMlp:
        conv, bn = seq[0], seq[1]
        return _abi.PtxSlotMlp(conv_w=_ptr(conv.weight), conv_b=_ptr(conv.bias), bn_w=_ptr(bn.weight),
                               bn_b=_ptr(bn.bias), bn_mean=_ptr(bn.running_mean), bn_var=_ptr(bn.running_var))

    @staticmethod
    def _bn_struct(bn: nn.BatchNorm1d) -> _abi.PtxBn1d:
        return _abi.PtxBn1d(w=_ptr(bn.weight), b=_ptr(bn.bias), mean=_ptr(bn.running_mean),
                            var=_ptr(bn.running_var))

    def _ensure_prepared(self, shape: _abi.PtxShape, device: torch.device, stream: int):
        """(Re)build the weight-pointer struct and the parameter-only tables when any
        parameter storage or version changed (load_state_dict, .to(), optimiser step ...)."""
        key = (self._weights_key(), str(device), shape.Mk)
        if key == self._wkey:
            return
        for name, t in self.state_dict(keep_vars=True).items():
            if t.device != device:
                raise RuntimeError(f"parameter {name} is on {t.device}, inputs are on {device}")
            if t.is_floating_point() and (t.dtype != torch.float32 or not t.is_contiguous()):
                raise RuntimeError(f"parameter {name} must be contiguous float32 (got {t.dtype})")
        ap = self.attn_pool2d
        # only the last block of each list reaches the output (PRE:441-443, 450-452)
        w = _abi.PtxWeights(
            offset=self._slot_struct(self.get_deformable_cluster.get_offsets.mlp),
            offset_map_w=_ptr(self.get_deformable_cluster.get_offsets.channel_mapper.weight),
            encoder=self._slot_struct(self.simple_encoder.mlp),
            cm_w=_ptr(self.channel_mapper.weight), cm_b=_ptr(self.channel_mapper.bias),
            pos=_ptr(ap.positional_embedding),
            q_w=_ptr(ap.q_proj.weight), q_b=_ptr(ap.q_proj.bias), k_w=_ptr(ap.k_proj.weight),
            k_b=_ptr(ap.k_proj.bias), v_w=_ptr(ap.v_proj.weight), v_b=_ptr(ap.v_proj.bias),
            c_w=_ptr(ap.c_proj.weight), c_b=_ptr(ap.c_proj.bias),
            norm_img_w=_ptr(self.norm_img.weight), norm_img_b=_ptr(self.norm_img.bias),
            text=self._block_struct(self.textformer[-1], self.text_norm[-1]),
            img=self._block_struct(self.imgformer[-1], self.img_norm[-1]),
            text_trans_w=_ptr(self.text_trans.weight), text_trans_b=_ptr(self.text_trans.bias),
            img_trans_w=_ptr(self.img_trans.weight), img_trans_b=_ptr(self.img_trans.bias),
            text_trans_norm=self._bn_struct(self.text_trans_norm),
            img_trans_norm=self._bn_struct(self.img_trans_norm))
        lib = _abi.lib()
        nbytes = lib.ptx_prep_bytes(ctypes.byref(shape))
        if nbytes == 0:
            raise RuntimeError("unsupported configuration: " + lib.ptx_last_error().decode())
        # forwards already enqueued on OTHER streams may still be reading the old tables (the lane of the calling
        # stream may not exist yet, so the test is per lane, not a count)
        for lane in list(self._lanes.values()):
            if lane.stream.cuda_stream != stream:
                lane.stream.synchronize()
        prep = torch.empty(nbytes, dtype=torch.uint8, device=device)
        # torch.linspace is part of the reference's arithmetic (PRE:41)
        lin = torch.linspace(0, 1, self.grid_size, device="cpu").to(device)
        _abi.check(lib.ptx_prepare(ctypes.byref(shape), ctypes.byref(w), lin.data_ptr(),
                                   prep.data_ptr(), nbytes, stream), "ptx_prepare")
        # ... and no stream may start on the new ones before they are built (weights change rarely in eval)
        torch.cuda.current_stream(device).synchronize()
        self._wstruct, self._prep, self._lin, self._wkey = w, prep, lin, key

    def _lane(self, device: torch.device, tstream) -> _Lane:
        """The call state of (device, stream), created on first use; at most ``_MAX_LANES`` are kept."""
        key = (str(device), tstream.cuda_stream)
        lane = self._lanes.get(key)
        if lane is None:
            if len(self._lanes) >= _MAX_LANES:
                old_key = next(iter(self._lanes))
                old = self._lanes.pop(old_key)
                old.stream.synchronize()           # its workspace may still be in use
                # a join gate of the retired lane's last forward may have failed after that forward returned: the error must
                # outlive the context that holds it -- check() / close() / the exit hook report it
                err = old.sync_check()
                if err:
                    self.__dict__.setdefault("_pending_errors", []).append(err)
                old.release()
            lane = self._lanes[key] = _Lane(tstream)
        else:
            self._lanes[key] = self._lanes.pop(key)   # most recently used last
        return lane

    def _release_lanes(self):
        lanes = self.__dict__.get("_lanes") or {}
        self.__dict__["_lanes"] = {}
        for lane in lanes.values():
            lane.release()

    def check(self) -> None:
        """Raise if a stream gate of any forward this module has handed out failed (not in the reference: PRE runs on one stream).

        ``forward`` returns as soon as the per-scene lengths are known -- before the forward has drained -- so a JOIN gate that
        runs out of time afterwards (the outputs of that forward are NaN) cannot be reported by the call itself; it is reported
        by the next ``forward`` on the same stream, by ``check()`` (which drains the streams the module used), by ``close()``, when
        the module is garbage-collected (a warning) and at interpreter exit (exit code 70).  A loop that does not set
        ``sync_outputs`` should call ``check()`` once behind its last forward."""
        errs = self.__dict__.get("_pending_errors") or []     # failures found while a lane was retired (_lane)
        self.__dict__["_pending_errors"] = []
        errs = errs + [e for e in (lane.sync_check() for lane in list(self._lanes.values())) if e]
        if errs:
            raise RuntimeError("ptx_forward: " + " | ".join(errs))

    def close(self) -> None:
        """``check()``, then release the library contexts / workspaces of every stream this module was called on."""
        try:
            self.check()
        finally:
            self._release_lanes()

    def __del__(self):
        # A finaliser runs at any collection point, also inside a torch.cuda.graph capture, where the drain of check() /
        # ptx_context_destroy would invalidate the capture.  There it only POLLS the pinned error words
        # (ptx_context_check, no synchronise) and parks the lanes on a module-level list that the exit hook drains, checks and
        # releases; outside a capture it is check() (a drain that ptx_context_destroy would perform anyway) + release.
        try:
            capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
            if capturing:
                lib = _abi.lib()
                errs = list(self.__dict__.get("_pending_errors") or [])
                lanes = self.__dict__.get("_lanes") or {}
                self.__dict__["_lanes"] = {}
                for lane in lanes.values():
                    if lane.ctx is not None and lane.unchecked and lib.ptx_context_check(lane.ctx) != 0:
                        errs.append(lib.ptx_last_error().decode())
                        lane.unchecked = False
                    _ORPHAN_LANES.append(lane)
                if errs:
                    raise RuntimeError(" | ".join(errs))
            else:
                try:
                    self.check()
                finally:
                    self._release_lanes()
        except RuntimeError as e:
            import warnings
            warnings.warn(f"{type(self).__name__} collected with an unreported failure: {e}", RuntimeWarning)
        except Exception:          # interpreter shutdown: modules may already be torn down
            pass

    def _workspace(self, lane: _Lane, shape: _abi.PtxShape, device: torch.device, stream: int) -> torch.Tensor:
        key = (shape.B, shape.N, shape.L, shape.V, str(device))           # layout does not depend on img_dtype
        if lane.ws is None or lane.ws_key != key:
            nbytes = _abi.lib().ptx_workspace_bytes(ctypes.byref(shape))
            if nbytes == 0:
                raise RuntimeError("unsupported configuration: " + _abi.lib().ptx_last_error().decode())
            lane.ws = None                          # one live workspace per lane
            lane.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            lane.ws_key, lane.ws_dirty = key, True
        if lane.ws_dirty:
            # tags / encoded boxes / count accumulators: zero once, the kernels keep them zero (no per-call memset)
            _abi.check(_abi.lib().ptx_workspace_init(ctypes.byref(shape), lane.ws.data_ptr(), lane.ws.numel(), stream),
                       "ptx_workspace_init")
            lane.ws_dirty = False
        return lane.ws

    # ------------------------------------------------------------------ forward
    def _check_inputs(self, points, text_dict, img_feat):
        """Validate the reference's input contract; returns device-ready tensors.

        The point clouds are used IN PLACE through a table of per-scene pointers (the path never
        writes to its input; the reference's stacked copy, PRE:426-427, exists only because its
        scatter is in-place).  Only irregular inputs (non-fp32, non-contiguous, > 32 scenes)
        are stacked into a fresh tensor."""
        if not isinstance(points, (list, tuple)) or len(points) == 0:
            raise ValueError("points must be a non-empty list of (N,3) tensors")
        p0 = points[0]
        if not p0.is_cuda:
            raise RuntimeError("ProxyTransformationNormReverse (HIP) needs GPU tensors: there is no CPU path")
        B, shp = len(points), p0.shape
        if len(shp) != 2 or shp[1] != 3:
            raise RuntimeError(f"points must be (N,3) per scene, got {tuple(shp)}")
        direct = B <= _MAX_SCENES_PER_CALL
        for p in points:
            if p.shape != shp:                                   # like torch.cat at PRE:427
                raise RuntimeError(f"all scenes must have the same number of points: {tuple(p.shape)} vs {tuple(shp)}")
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != p0.device:
                direct = False
        if direct:
            pts, plist = None, (ctypes.c_void_p * B)(*[p.data_ptr() for p in points])
        else:
            pts, plist = torch.stack([p.to(device=p0.device, dtype=torch.float32) for p in points]).contiguous(), None
        text_feats, text_mask = self.get_text_proxy(text_dict)   # positional unpack (PRE:440)
        for name, tns in (("text_feats", text_feats), ("text_token_mask", text_mask), ("img_feat", img_feat)):
            if not isinstance(tns, torch.Tensor) or tns.device != p0.device:
                raise RuntimeError(f"{name} must be a tensor on {p0.device} (the device of points), got "
                                   f"{getattr(tns, 'device', type(tns))}")
        if text_feats.shape[0] != B or text_feats.shape[-1] != self.embed_dim or text_feats.dim() != 3:
            raise RuntimeError(f"text_feats must be ({B},L,{self.embed_dim}), got {tuple(text_feats.shape)}")
        if text_mask.shape != text_feats.shape[:2]:
            raise RuntimeError("text_token_mask must be (B,L)")
        hw = self.img_spacial_dim
        ish = img_feat.shape
        if len(ish) != 5 or ish[0] != B or ish[2] != self.input_dim or ish[3] != hw or ish[4] != hw:
            raise RuntimeError(f"img_feat must be ({B},V,{self.input_dim},{hw},{hw}), got {tuple(ish)}")
        assert self.real_cluster_num >= 1                        # PRE:209
        if text_feats.dtype != torch.float32 or not text_feats.is_contiguous():
            text_feats = text_feats.to(torch.float32).contiguous()
        if text_mask.dtype == torch.bool and text_mask.is_contiguous():
            mask_u8 = text_mask.view(torch.uint8)                 # zero-copy: bool is one byte, 0 / 1
        else:
            mask_u8 = (text_mask != 0).to(torch.uint8).contiguous()
        # image features are consumed in their storage type (fp32, or bf16 / fp16 from an AMP backbone);
        # arithmetic is fp32 either way
        if img_feat.dtype not in _IMG_DTYPES:
            img_feat = img_feat.to(torch.float32)
        if not img_feat.is_contiguous():
            img_feat = img_feat.contiguous()
        return (B, shp[0], p0.device), pts, plist, text_feats, mask_u8, img_feat

    def _run(self, points, text_dict, img_feat, debug: bool, transforms: bool = False, bbox=None):
        (B, N, dev), pts, plist, text_feats, mask_u8, img = self._check_inputs(points, text_dict, img_feat)
        skey = (B, N, text_feats.shape[1], img.shape[1], _IMG_DTYPES[img.dtype])
        shape = self._shapes.get(skey)
        if shape is None:
            shape = self._shapes[skey] = self._shape(*skey)
        lib = _abi.lib()
        if dev.index is not None and dev.index != torch.cuda.current_device():
            raise RuntimeError(f"inputs are on {dev} but the current device is cuda:{torch.cuda.current_device()}")
        tstream = torch.cuda.current_stream(dev)
        stream = tstream.cuda_stream
        self._ensure_prepared(shape, dev, stream)
        lane = self._lane(dev, tstream)
        ws = self._workspace(lane, shape, dev, stream)
        out = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        # per-scene survivor counts land directly in pinned (device-mapped) host memory, published
        # by the clustering chain as soon as the drop tags are final: the host only waits for
        # those B integers (the list lengths of PRE:467), not for the forward to drain
        if lane.counts is None or lane.counts.numel() < B:
            lane.counts = torch.empty((max(B, 64),), dtype=torch.int32).pin_memory()
            lane.counts_np = lane.counts.numpy()
        counts = lane.counts
        lane.counts_np[:B] = -1
        dbg_struct, dbg = None, {}
        if debug:
            dbg = self._alloc_debug(shape, dev)
        elif transforms:
            # the per-cluster transforms only: three small stream-ordered copies, no drain
            dbg = self._alloc_debug(shape, dev, only=("kcenter", "translate", "transform"))
        if dbg:
            dbg_struct = _abi.PtxDebug(**{k: v.data_ptr() for k, v in dbg.items()})
        oo = self._order_override
        co = self._centers_override
        if oo is not None:
            oo = oo.to(device=dev, dtype=torch.int32).contiguous()
        if co is not None:
            co = co.to(device=dev, dtype=torch.float32).contiguous()
        opts = None
        if bbox is not None or self.compute_dtype != "fp32":
            if bbox is not None and (not isinstance(bbox, torch.Tensor) or bbox.device != dev or bbox.shape != (B, 6)
                                     or bbox.dtype != torch.int32 or not bbox.is_contiguous()):
                raise RuntimeError(f"bbox must be a contiguous ({B},6) int32 tensor on {dev} (IngestedBatch.bbox)")
            opts = _abi.PtxForwardOpts(bbox_enc=_ptr(bbox), compute_dtype=_COMPUTE_DTYPES[self.compute_dtype])
        lane.ws_dirty = True              # until the call has been enqueued completely
        _abi.check(lib.ptx_forward_ex(
            lane.ctx, ctypes.byref(shape), ctypes.byref(self._wstruct), self._prep.data_ptr(),
            self._lin.data_ptr(), _ptr(pts), plist, text_feats.data_ptr(), mask_u8.data_ptr(),
            img.data_ptr(), _ptr(oo), _ptr(co), out.data_ptr(), counts.data_ptr(),
            ws.data_ptr(), ws.numel(), ctypes.byref(dbg_struct) if dbg_struct else None,
            ctypes.byref(opts) if opts is not None else None, stream),
            "ptx_forward")
        lane.ws_dirty = False
        drained = False
        if debug or self.sync_outputs or lib.ptx_wait_counts(counts.data_ptr(), B, _COUNTS_TIMEOUT_US) != 0:
            tstream.synchronize()                          # full drain; also surfaces device faults
            drained = True
        # a stream gate that ran out of time (csrc/api.hip, "gates") has stored its error word by now if it was the fork; a join
        # that fails later (after the counts) turns this call's outputs into NaN and is reported by the next forward on this
        # stream, by check() / close(), at garbage collection and at interpreter exit -- or here, when the call drained the stream
        lane.unchecked = not drained and lib.ptx_context_gates(lane.ctx) != 0
        if lib.ptx_context_check(lane.ctx) != 0:
            lane.ws_dirty = True                           # the workspace's clean-on-entry words cannot be trusted
            raise RuntimeError("ptx_forward: " + lib.ptx_last_error().decode())
        n_keep = lane.counts_np[:B].tolist()
        if min(n_keep) < 0:
            raise RuntimeError("ptx_forward finished without publishing the survivor counts")
        outs = [out[b, :n_keep[b]] for b in range(B)]
        return outs, dbg

    @torch.no_grad()
    def forward_padded(self, points: List[torch.Tensor], text_dict: dict, img_feat: torch.Tensor,
                       bbox: Optional[torch.Tensor] = None):
        """The eval forward WITHOUT its one host wait (not in the reference): returns ``(out, counts)`` -- ``out`` (B,N,3), scene b's
        transformed, compacted points in ``out[b, :counts[b]]`` (rows beyond are unspecified), ``counts`` (B,) int32 on the device --
        both ordered on the current stream.  ``forward`` is this plus ``counts`` read on the host and the list of views.

        Nothing here synchronises or allocates outside torch's allocator, so the call can be captured into a HIP graph
        (``torch.cuda.graph``): the library's side streams fork from and join into the capturing stream through events.
        Warm up first on the stream the capture will use (``torch.cuda.graph(g, stream=s)``): the lane of a stream -- library
        context, workspace, parameter tables -- is created on its first call, which allocates and synchronises."""
        if self.training:
            raise RuntimeError("forward_padded is the eval path; train mode returns ragged outputs through forward()")
        (B, N, dev), pts, plist, text_feats, mask_u8, img = self._check_inputs(points, text_dict, img_feat)
        if B > _MAX_SCENES_PER_CALL:
            raise RuntimeError(f"forward_padded takes at most {_MAX_SCENES_PER_CALL} scenes per call (got {B})")
        skey = (B, N, text_feats.shape[1], img.shape[1], _IMG_DTYPES[img.dtype])
        shape = self._shapes.get(skey)
        if shape is None:
            shape = self._shapes[skey] = self._shape(*skey)
        lib = _abi.lib()
        tstream = torch.cuda.current_stream(dev)
        stream = tstream.cuda_stream
        capturing = torch.cuda.is_current_stream_capturing()
        key = (str(dev), stream)
        if capturing and (key not in self._lanes or self._wkey is None or self._lanes[key].ws is None or self._lanes[key].ws_dirty):
            raise RuntimeError("forward_padded inside a stream capture: call it once on this stream BEFORE capturing (the lane's "
                               "context / workspace / parameter tables are created on the first call)")
        if not capturing:
            self._ensure_prepared(shape, dev, stream)
        lane = self._lane(dev, tstream)
        ws = self._workspace(lane, shape, dev, stream)
        out = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        opts = None
        if bbox is not None or self.compute_dtype != "fp32":
            opts = _abi.PtxForwardOpts(bbox_enc=_ptr(bbox), compute_dtype=_COMPUTE_DTYPES[self.compute_dtype])
        oo, co = self._order_override, self._centers_override
        if oo is not None or co is not None:
            raise RuntimeError("forward_padded does not take the test-only overrides")
        lane.ws_dirty = True
        _abi.check(lib.ptx_forward_ex(
            lane.ctx, ctypes.byref(shape), ctypes.byref(self._wstruct), self._prep.data_ptr(),
            self._lin.data_ptr(), _ptr(pts), plist, text_feats.data_ptr(), mask_u8.data_ptr(),
            img.data_ptr(), None, None, out.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(), None,
            ctypes.byref(opts) if opts is not None else None, stream), "ptx_forward")
        lane.ws_dirty = False
        lane.unchecked = not capturing and lib.ptx_context_gates(lane.ctx) != 0     # nothing was waited for: check() is the caller's
        if capturing:
            # the captured kernels read these through raw pointers: they have to outlive the graph
            keep = self.__dict__.get("_graph_keepalive") or []
            keep.append((pts, plist, text_feats, mask_u8, img, self._prep, self._lin, ws))
            self._graph_keepalive = keep
        return out, counts

    def _alloc_debug(self, s: _abi.PtxShape, dev, only=None) -> Dict[str, torch.Tensor]:
        M, K, Kd = self.num_cluster, s.K, s.Mt - s.Mk
        f32, i32 = torch.float32, torch.int32
        spec = dict(
            centers0=((s.B, M, 3), f32), cluster1=((s.B, M, K, 3), f32), offsets=((s.B, M, 3), f32),
            centers=((s.B, M, 3), f32), cluster2=((s.B, M, K, 3), f32), idx2=((s.B, M, K), i32),
            pad_count=((s.B, M), i32), order=((s.B, s.Mt), i32), picks=((s.B, Kd), i32),
            keep=((s.B, s.Mk), i32), kidx=((s.B, s.Mk, K), i32), drop_idx=((s.B, Kd * K), i32),
            kcenter=((s.B, s.Mk, 3), f32), kcluster=((s.B, s.Mk, K, 3), f32),
            point_proxy=((s.B, s.Mk, s.C), f32), img_proxy=((s.B, s.V, s.C), f32),
            text_guide=((s.B, s.Mk, s.C), f32), img_guide=((s.B, s.Mk, s.C), f32),
            translate=((s.B, s.Mk, 3), f32), transform=((s.B, s.Mk, 9), f32),
            tag=((s.B, s.N), torch.int32))
        return {k: torch.empty(shp, dtype=dt, device=dev) for k, (shp, dt) in spec.items()
                if only is None or k in only}

    def forward(self, points: List[torch.Tensor], text_dict: dict, img_feat: torch.Tensor,
                return_transforms: bool = False, bbox: Optional[torch.Tensor] = None):
        """points: list of B (N,3) fp32 GPU tensors; text_dict.values() -> (text_feats (B,L,C),
        text_token_mask (B,L) bool, True = valid); img_feat (B,V,input_dim,H,W).
        Returns a list of B tensors (N_i',3): transformed points, dropped points removed,
        original order preserved (PRE:424-469).

        ``return_transforms=True`` (not in the reference) additionally returns the per-cluster affine
        parameters ``dict(kcenter (B,M',3), translate (B,M',3), transform (B,M',9))`` -- what
        ``shard.gather_cluster_transforms`` exchanges between ranks -- as stream-ordered tensors.

        ``bbox`` (not in the reference): ``IngestedBatch.bbox`` of ``ingest.MultiViewIngest`` -- the clouds' encoded
        bounding boxes, reduced while the points were written; the eval forward then skips its own min / max pass over
        the points (PRE:37-38).  Train mode computes its boxes itself."""
        if self.training:
            if return_transforms:
                outs, aux = self._run_train(points, text_dict, img_feat)
                return outs, {k: aux[k].view(len(points), self.real_cluster_num, -1)
                              for k in ("kcenter", "translate", "transform")}
            return self._run_train(points, text_dict, img_feat)[0]
        if torch.is_grad_enabled() and not self._warned_eval_grad and any(p.requires_grad for p in self.parameters()):
            import warnings
            self._warned_eval_grad = True
            warnings.warn("ProxyTransformationNormReverse in eval mode returns tensors without grad_fn (the eval path is "
                          "inference-only HIP); call .train() to differentiate, or wrap the call in torch.no_grad()")
        chunks = [(0, len(points))]
        if isinstance(points, (list, tuple)) and len(points) > _MAX_SCENES_PER_CALL:
            # scenes are independent in eval mode: larger batches run as consecutive calls
            chunks = [(i, min(i + _MAX_SCENES_PER_CALL, len(points)))
                      for i in range(0, len(points), _MAX_SCENES_PER_CALL)]
        if len(chunks) == 1:
            outs, extra = self._run(points, text_dict, img_feat, debug=False, transforms=return_transforms, bbox=bbox)
            return (outs, extra) if return_transforms else outs
        feats, mask = self.get_text_proxy(text_dict)
        outs: List[torch.Tensor] = []
        extras = []
        for i, j in chunks:
            o, e = self._run(points[i:j], {"text_feats": feats[i:j], "text_token_mask": mask[i:j]},
                             img_feat[i:j], debug=False, transforms=return_transforms,
                             bbox=None if bbox is None else bbox[i:j])
            outs += o
            extras.append(e)
        if return_transforms:
            return outs, {k: torch.cat([e[k] for e in extras]) for k in extras[0]}
        return outs

    def _run_train(self, points, text_dict, img_feat):
        """Train-mode forward (batch-statistics BatchNorm, Dropout / DropPath, autograd graph of HIP kernels):
        see ``train.py``.  One scene list of at most 32 scenes per call (the reference trains with 6, CFG:145)."""
        from . import train
        (B, N, dev), pts, plist, text_feats, mask_u8, img = self._check_inputs(points, text_dict, img_feat)
        if B > _MAX_SCENES_PER_CALL:
            raise RuntimeError(f"train mode takes at most {_MAX_SCENES_PER_CALL} scenes per call (got {B})")
        # identity of every live parameter / buffer OBJECT, read from the owning dicts on every step (~10 us): a Parameter swapped in
        # without load_state_dict / .to() / train() -- a reparametrisation, convert_sync_batchnorm after the first step -- must not
        # keep receiving `grad None` through a stale gradient list, nor skip the layout / BatchNorm checks below
        # a swapped sub-MODULE (convert_sync_batchnorm) owns new dicts: every (owner's _modules dict, name, child) edge of the tree
        # is compared by identity -- walking self.modules() itself cost 0.1 ms of a host-bound 2 ms step (r05)
        edges = getattr(self, "_train_mods", None)
        if edges is None or not all([d.get(k) is c for d, k, c in edges]):
            self._train_mods = [(m._modules, k, c) for m in self.modules() for k, c in m._modules.items()]
            self.invalidate_weights()
        if self._slots is None:
            self._weights_key()
        live_ids = tuple([id(d[k]) for d, k in self._slots])
        if self._train_checked != (str(dev), live_ids):
            self._train_live = self._train_static = None
            # layout of every parameter / buffer: once per (device, storage generation) -- invalidate_weights() (load_state_dict,
            # .to(), train() / eval()) asks for it again; walking the state_dict on every step cost 0.1 ms
            for name, t in self.state_dict(keep_vars=True).items():
                if t.device != dev or (t.is_floating_point() and (t.dtype != torch.float32 or not t.is_contiguous())):
                    raise RuntimeError(f"parameter {name} must be contiguous float32 on {dev}")
            for name, bn in self._batch_norms():
                if type(bn) not in (nn.BatchNorm1d, nn.BatchNorm2d):
                    raise NotImplementedError(f"{name} is a {type(bn).__name__}: the HIP train path computes local batch "
                                              "statistics (the reference trains with plain DDP, no SyncBatchNorm)")
                if bn.momentum is None:
                    raise NotImplementedError(f"{name}.momentum=None (cumulative moving average) is not implemented")
            self._train_checked = (str(dev), live_ids)
        if dev.index is not None and dev.index != torch.cuda.current_device():
            raise RuntimeError(f"inputs are on {dev} but the current device is cuda:{torch.cuda.current_device()}")
        shape = self._shape(B, N, text_feats.shape[1], img.shape[1], _IMG_DTYPES[img.dtype])
        tstream = torch.cuda.current_stream(dev)
        lane = self._lane(dev, tstream)
        ws = self._workspace(lane, shape, dev, tstream.cuda_stream)
        # _check_inputs hands back the caller's own tensors when their layout is already right, so gradients w.r.t.
        # text_feats / img_feat reach them directly
        tf, im = text_feats, img
        res = train.forward_train(self, list(points), tf, mask_u8, im, shape, ws, self._order_override)
        # the running statistics were updated through raw pointers: tell torch (and _weights_key) that they changed
        bns = [bn for _, bn in self._batch_norms()]
        for bn in bns:                                     # no kernel: only the version counters move
            torch.autograd.graph.increment_version(bn.running_mean)
            torch.autograd.graph.increment_version(bn.running_var)
        tracked = [bn.num_batches_tracked for bn in bns if bn.num_batches_tracked is not None]
        if tracked:
            torch._foreach_add_(tracked, 1)                # one launch for the four counters
        return res

    def uses_stream_gates(self, stream=None) -> bool:
        """True when the lane of ``stream`` (default: the current one) orders its two chains with device-word gates
        rather than events (after its first forward: the concurrency probe runs there)."""
        dev = next(self.parameters()).device
        tstream = stream if stream is not None else torch.cuda.current_stream(dev)
        lane = self._lanes.get((str(dev), tstream.cuda_stream))
        return bool(lane is not None and _abi.lib().ptx_context_gates(lane.ctx))

    def stream_gate_bits(self, stream=None) -> int:
        """``ptx_context_gates`` of the lane: bit 0 gates in use, bit 1 the low-priority stream may carry gate words too (the
        clustering-chain layout's join / tags words need both)."""
        dev = next(self.parameters()).device
        tstream = stream if stream is not None else torch.cuda.current_stream(dev)
        lane = self._lanes.get((str(dev), tstream.cuda_stream))
        return int(_abi.lib().ptx_context_gates(lane.ctx)) if lane is not None else 0

    def _batch_norms(self):
        return (("get_deformable_cluster.get_offsets.mlp.1", self.get_deformable_cluster.get_offsets.mlp[1]),
                ("simple_encoder.mlp.1", self.simple_encoder.mlp[1]),
                ("text_trans_norm", self.text_trans_norm), ("img_trans_norm", self.img_trans_norm))

    @torch.no_grad()
    def quantize(self, outs: List[torch.Tensor], voxel_size: float = 0.01, return_inverse: bool = False,
                 return_scene_rows: bool = False):
        """What the reference's detector does with this module's output next (detectors/
        sparse_featfusion_grounder_preshape.py:388-397, ``use_xyz_feat``): ``ME.utils.batch_sparse_collate([(p / voxel_size,
        p) ...])`` + ``ME.SparseTensor`` -- coordinates ``(Nv,4) int32 = (scene, floor(p / voxel_size))`` and features
        ``(Nv,3)``, one row per occupied voxel (the first point of a voxel in (scene, point) order; MinkowskiEngine's own
        choice is unspecified).  ``outs`` = the list ``forward`` returned (views of one padded buffer are used in place,
        anything else is packed).  Runs on the current stream; the host waits only for the row count, which the last kernel
        publishes through pinned memory (the tensors' contents are stream-ordered like any torch result).

        ``return_scene_rows=True`` appends ``ends`` (list of B ints): the rows of scene b are ``[ends[b-1], ends[b])`` -- what
        ``x.decomposed_coordinates`` (DET:391-392, 429-430) splits by; published through the same pinned words as the count."""
        lib = _abi.lib()
        B = len(outs)
        dev = outs[0].device
        n = [int(o.shape[0]) for o in outs]
        base = outs[0]._base if outs[0]._base is not None else None
        packed = base is not None and base.dim() == 3 and base.shape[0] == B and base.dtype == torch.float32 and all(
            o._base is base and o.data_ptr() == base[b].data_ptr() for b, o in enumerate(outs))
        if packed:
            buf, Ncap = base, base.shape[1]
        else:
            Ncap = max(max(n), 1)
            buf = torch.zeros((B, Ncap, 3), dtype=torch.float32, device=dev)
            for b, o in enumerate(outs):
                buf[b, : n[b]].copy_(o)
        tstream = torch.cuda.current_stream(dev)
        lane = self._lane(dev, tstream)
        # the scratch of this step lives with the lane (one allocation per shape, not five per call); the row count comes
        # back through pinned memory as soon as the rows are written -- no stream drain, no device-to-host copy
        qkey = (B, Ncap, str(dev))
        q = lane.quant if lane.quant is not None and lane.quant["key"] == qkey else None
        if q is None:
            nbytes = lib.ptx_voxel_workspace_bytes(B, Ncap)
            if nbytes == 0:
                raise RuntimeError(f"quantize: unsupported size B={B}, N={Ncap}")
            q = lane.quant = dict(key=qkey, ws=torch.empty((nbytes,), dtype=torch.uint8, device=dev),
                                  counts_h=torch.empty((max(B, 1),), dtype=torch.int32).pin_memory(),
                                  info=torch.empty((2 + max(B, 1),), dtype=torch.int32).pin_memory())
            q["info_np"] = q["info"].numpy()
        q["counts_h"][:B] = torch.tensor(n, dtype=torch.int32)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        counts.copy_(q["counts_h"][:B], non_blocking=True)       # the staging buffer is free again once `info` is published
        coords = torch.empty((B * Ncap, 4), dtype=torch.int32, device=dev)
        feats = torch.empty((B * Ncap, 3), dtype=torch.float32, device=dev)
        inverse = torch.empty((B, Ncap), dtype=torch.int32, device=dev) if return_inverse else None
        q["info_np"][:] = -1
        _abi.check(lib.ptx_voxelize_ex(buf.data_ptr(), counts.data_ptr(), B, Ncap, float(voxel_size), coords.data_ptr(),
                                       feats.data_ptr(), _ptr(inverse), q["info"].data_ptr(), q["info"].data_ptr() + 8,
                                       q["ws"].data_ptr(), q["ws"].numel(), tstream.cuda_stream), "ptx_voxelize")
        if lib.ptx_wait_counts(q["info"].data_ptr(), 2 + B, _COUNTS_TIMEOUT_US) != 0:
            tstream.synchronize()
        nvox, overflow = (int(x) for x in q["info_np"][:2])
        ends = q["info_np"][2:2 + B].tolist()
        if nvox < 0:
            raise RuntimeError("ptx_voxelize finished without publishing its row count")
        if nvox == 0x7fffffff:           # PTX_VOX_BROKEN: a tile of the emit pass gave up waiting for the tiles in front of it
            raise RuntimeError("ptx_voxelize: the single-pass scan of the emit kernel timed out (a tile never published its count); "
                               "rows, inverse and the row count of this call are invalid")
        # (the rows themselves -- coords / feats / inverse -- are ordered on the stream like any other result; only the COUNT is known
        #  here: a consumer on another stream has to wait for this stream, as for any torch tensor)
        if overflow:
            raise RuntimeError(f"quantize: {overflow} points fall outside +-2^18 voxels of size {voxel_size}")
        res = (coords[:nvox], feats[:nvox])
        if return_inverse:
            res += ([inverse[b, : n[b]] for b in range(B)],)
        if return_scene_rows:
            res += (ends,)
        return res

    @torch.no_grad()
    def forward_debug(self, points, text_dict, img_feat, bbox=None):
        """forward + every intermediate the C ABI can export (tests / parity only)."""
        outs, dbg = self._run(points, text_dict, img_feat, debug=True, bbox=bbox)
        dbg["outputs"] = outs
        return dbg
