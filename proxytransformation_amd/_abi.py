"""ctypes binding of ``libproxyt_hip.so`` (C ABI declared in ``include/proxyt.h``).

The library is the product: there is no CPU or PyTorch fallback.  Importing this
module never touches the GPU; :func:`lib` raises ``RuntimeError`` with build
instructions if the shared object is missing, and every ``ptx_*`` call that
returns a negative code raises ``RuntimeError`` carrying ``ptx_last_error()``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library.  The in-tree build with the stream gates' fault-injection hooks compiled in (csrc/Makefile:
# libproxyt_hip_testhooks.so) is reachable only through the explicit test-only call ``use_test_hooks_library()`` below, made by
# tests/test_gpu_host.py's gate-failure workers BEFORE anything is loaded -- no environment variable can swap the library of a
# process.  Nothing else can be loaded.
LIB_PATH = os.path.join(_HERE, "libproxyt_hip.so")
ABI_VERSION = 12

c_float_p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())


class PtxShape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "N", "grid_size", "K", "Mt", "Mk", "L", "V", "C", "heads", "hidden",
                 "in_dim", "hw", "img_dtype")] + \
               [(n, C.c_float) for n in ("radius", "margin", "bn_eps", "ln_eps")]


class PtxSlotMlp(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("conv_w", "conv_b", "bn_w", "bn_b", "bn_mean", "bn_var")]


class PtxBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("norm1_w", "norm1_b", "pb_bias", "pc_bias", "pr_bias", "qkv_w", "qkv_b",
                 "pp_w", "pp_b", "proj_w", "proj_b", "norm2_w", "norm2_b",
                 "fc1_w", "fc1_b", "fc2_w", "fc2_b", "out_norm_w", "out_norm_b")]


class PtxBn1d(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w", "b", "mean", "var")]


class PtxWeights(C.Structure):
    _fields_ = [("offset", PtxSlotMlp), ("offset_map_w", C.c_void_p), ("encoder", PtxSlotMlp),
                ("cm_w", C.c_void_p), ("cm_b", C.c_void_p), ("pos", C.c_void_p)] + \
               [(n, C.c_void_p) for n in ("q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "c_w", "c_b",
                                          "norm_img_w", "norm_img_b")] + \
               [("text", PtxBlock), ("img", PtxBlock)] + \
               [(n, C.c_void_p) for n in ("text_trans_w", "text_trans_b", "img_trans_w", "img_trans_b")] + \
               [("text_trans_norm", PtxBn1d), ("img_trans_norm", PtxBn1d)]


DEBUG_FIELDS = ("centers0", "cluster1", "offsets", "centers", "cluster2",
                "idx2", "pad_count", "order", "picks", "keep", "kidx", "drop_idx",
                "kcenter", "kcluster", "point_proxy", "img_proxy", "text_guide", "img_guide",
                "translate", "transform", "tag")


class PtxDebug(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in DEBUG_FIELDS]


TB_PARAMS = ("ln1_w", "ln1_b", "pb", "pc", "pr", "qkv_w", "qkv_b", "pp_w", "pp_b", "proj_w", "proj_b", "ln2_w", "ln2_b",
             "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ln3_w", "ln3_b", "head_w", "head_b", "bn_w", "bn_b")      # PTX_TB_*


class PtxTrainBlock(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "n", "L", "C", "H", "heads", "s", "nout", "compute_dtype")] + \
               [(n, C.c_float) for n in ("eps1", "eps2", "eps3", "bn_eps", "bn_momentum", "p_attn", "p_drop", "p_path")] + \
               [("seed", C.c_uint64 * 6), ("x", C.c_void_p), ("proxy", C.c_void_p), ("mask", C.c_void_p),
                ("param", C.c_void_p * len(TB_PARAMS)), ("bn_run_mean", C.c_void_p), ("bn_run_var", C.c_void_p),
                ("out", C.c_void_p), ("save", C.c_void_p), ("save_floats", C.c_size_t), ("tmp", C.c_void_p),
                ("tmp_floats", C.c_size_t), ("dout", C.c_void_p), ("dx", C.c_void_p), ("dproxy", C.c_void_p),
                ("grad", C.c_void_p * len(TB_PARAMS)), ("dx_add", C.c_void_p)]


class PtxTrainImgPool(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nimg", "Cin", "hw", "C", "heads", "img_dtype")] + \
               [(n, C.c_void_p) for n in ("img", "wc", "bc", "pos", "wq", "bq", "wk", "bk", "wv", "bv", "o", "save")] + \
               [("save_floats", C.c_size_t), ("tmp", C.c_void_p), ("tmp_floats", C.c_size_t)] + \
               [(n, C.c_void_p) for n in ("dout", "dimg", "dwc", "dbc", "dpos", "dwq", "dbq", "dwk", "dbk", "dwv", "dbv")] + \
               [(n, C.c_void_p) for n in ("cw", "cb", "lnw", "lnb")] + [("ln_eps", C.c_float)] + \
               [(n, C.c_void_p) for n in ("proxy", "dproxy", "dcw", "dcb", "dlnw", "dlnb")]


class PtxTrainSlotNet(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("conv_w", "conv_b", "bn_w", "bn_b", "run_mean", "run_var")] + \
               [("eps", C.c_float), ("momentum", C.c_float), ("W", C.c_int32)]


TS_NGRAD = 9 + 13 + 2 * len(TB_PARAMS)           # PTX_TS_NGRAD
TS_IP0, TS_TB0, TS_IB0 = 9, 9 + 13, 9 + 13 + len(TB_PARAMS)


class PtxTrainStepLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("arena_fwd_bytes", "arena_bwd_bytes", "grads_floats", "idx2", "order", "picks", "keep", "kidx",
                                           "drop_idx", "centers", "translate", "transform", "point_proxy", "img_proxy", "kcenter",
                                           "opos")] + [("grad_off", C.c_int64 * TS_NGRAD)]


class PtxTrainStep(C.Structure):
    _fields_ = [("shape", PtxShape), ("points", C.c_void_p), ("lin", C.c_void_p), ("centers_override", C.c_void_p),
                ("order_override", C.c_void_p), ("text_feats", C.c_void_p), ("text_mask", C.c_void_p),
                ("off", PtxTrainSlotNet), ("enc", PtxTrainSlotNet), ("map_w", C.c_void_p),
                ("ip", PtxTrainImgPool), ("tb", PtxTrainBlock), ("ib", PtxTrainBlock),
                ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("out", C.c_void_p), ("counts_host", C.c_void_p),
                ("arena_fwd", C.c_void_p), ("arena_fwd_bytes", C.c_size_t),
                ("side_stream", C.c_void_p), ("ev_fork", C.c_void_p), ("ev_join", C.c_void_p), ("ev_pp", C.c_void_p),
                ("ev_counts", C.c_void_p), ("blocks_apart", C.c_int32),
                ("douts", C.c_void_p), ("g_kcenter", C.c_void_p), ("g_translate", C.c_void_p), ("g_transform", C.c_void_p),
                ("arena_bwd", C.c_void_p), ("arena_bwd_bytes", C.c_size_t), ("grads", C.c_void_p), ("grads_floats", C.c_size_t),
                ("dtext", C.c_void_p), ("dimg", C.c_void_p)]


class PtxForwardOpts(C.Structure):
    _fields_ = [("bbox_enc", C.c_void_p), ("compute_dtype", C.c_int32), ("reserved", C.c_int32 * 5)]


# name -> (restype, argtypes); must list every symbol declared in include/proxyt.h
_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_SH, _W = C.POINTER(PtxShape), C.POINTER(PtxWeights)
SIGNATURES = {
    "ptx_abi_version": (C.c_int, []),
    "ptx_last_error": (C.c_char_p, []),
    "ptx_kernel_count": (C.c_int, []),
    "ptx_kernel_name": (C.c_char_p, [_I]),
    "ptx_timing_select": (_I, [_I]),
    "ptx_timing_every": (_I, [_I]),
    "ptx_timing_read": (_I, [C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "ptx_timing_select_mask": (_I, [C.c_uint64]),
    "ptx_timing_read_sites": (_I, [C.POINTER(C.c_int), C.POINTER(C.c_float), _I]),
    "ptx_prep_bytes": (_Z, [_SH]),
    "ptx_workspace_bytes": (_Z, [_SH]),
    "ptx_workspace_init": (_I, [_SH, _P, _Z, _P]),
    "ptx_prepare": (_I, [_SH, _W, _P, _P, _Z, _P]),
    "ptx_grid_centers": (_I, [_P, _I, _I, _P, _I, _F, _P, _P, _P, _Z, _P]),
    "ptx_ball_query": (_I, [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "ptx_linear": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ptx_gemm_policy": (_I, [_I]),
    "ptx_offset_net": (_I, [_SH, _W, _P, _P, _P, _P, _P, _P, _P]),
    "ptx_select_clusters": (_I, [_SH] + [_P] * 14),
    "ptx_pointnet": (_I, [_SH, _W, _P, _P, _P, _P, _P]),
    "ptx_img_proxy": (_I, [_SH, _W, _P, _P, _P, _P, _Z, _P]),
    "ptx_proxy_block": (_I, [_SH, _W, _P, _I, _P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "ptx_affine_scatter": (_I, [_SH, _P, _P, _P, _P, _P, _P, _P]),
    "ptx_affine_compact": (_I, [_SH, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "ptx_wait_counts": (_I, [_P, _I, C.c_int64]),
    "ptx_context_create": (_I, [C.POINTER(C.c_void_p)]),
    "ptx_context_destroy": (_I, [_P]),
    "ptx_context_check": (_I, [_P]),
    "ptx_context_sync_check": (_I, [_P]),
    "ptx_context_gates": (_I, [_P]),
    "ptx_forward": (_I, [_P, _SH, _W, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z,
                         C.POINTER(PtxDebug), _P]),
}

_L, _U64 = C.c_long, C.c_uint64
SIGNATURES.update({
    "ptx_forward_ex": (_I, [_P, _SH, _W, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z,
                            C.POINTER(PtxDebug), C.POINTER(PtxForwardOpts), _P]),
    "ptx_proxy_attention": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ptx_proxy_attention_scratch_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "ptx_ingest_workspace_bytes": (_Z, [_I, _I, _I]),
    "ptx_ingest_index": (_I, [_P, _I, _I, _I, _I, _P, _Z, _P, _P]),
    "ptx_ingest_gather": (_I, [_P, _I, _F, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "ptx_voxel_workspace_bytes": (_Z, [_I, _I]),
    "ptx_voxelize": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _Z, _P]),
    "ptx_voxelize_ex": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "ptx_voxel_coarsen": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _Z, _P]),
    "ptx_point_sample_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "ptx_point_sample_prepare": (_I, [_P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "ptx_point_sample": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _F, _F, _F, _F, _I, _F, _F, _F, _I, _P, _P, _P, _Z, _P]),
    "ptx_op_gemm": (_I, [_P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _L, _L, _I, _I, _L, _L, _L, _L, _L, _L, _I, _I, _F, _I, _I, _L, _P]),
    "ptx_op_transpose": (_I, [_P, _I, _I, _P, _P]),
    "ptx_op_colsum": (_I, [_P, _P, _I, _I, _I, _F, _I, _P, _P, _I, _P]),
    "ptx_op_eltwise": (_I, [_I, _P, _P, _F, _L, _I, _P, _P]),
    "ptx_op_dropout": (_I, [_P, _L, _L, _F, _U64, _P, _P]),
    "ptx_op_layernorm_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "ptx_op_layernorm_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "ptx_op_bn_stats": (_I, [_P, _P, _I, _L, _F, _F, _P, _P, _P, _P]),
    "ptx_op_bn_apply": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P]),
    "ptx_op_bn_bwd_prep": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "ptx_op_bn_bwd_dx": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "ptx_op_softmax_fwd": (_I, [_P, _P, _L, _I, _L, _P, _P]),
    "ptx_op_softmax_bwd": (_I, [_P, _P, _P, _L, _I, _L, _P, _P]),
    "ptx_op_slot_inputs": (_I, [_P, _P, _P, _L, _I, _P, _P, _P]),
    "ptx_op_slot_inputs_bwd": (_I, [_P, _P, _L, _I, _P, _P]),
    "ptx_op_slot_pool": (_I, [_P, _L, _I, _I, _I, _P, _P, _P]),
    "ptx_op_slot_pool_bwd": (_I, [_P, _P, _L, _I, _I, _I, _P, _P]),
    "ptx_op_slotnet_scratch_bytes": (_Z, [_I]),
    "ptx_op_slotnet_fwd": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "ptx_op_slotnet_bwd": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "ptx_op_offset_apply": (_I, [_P, _P, _P, _L, _I, _F, _P, _P, _P]),
    "ptx_op_slotbias_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "ptx_op_slotbias_bwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "ptx_op_keep_rows": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "ptx_op_rows_gather": (_I, [_P, _P, _L, _I, _P, _P]),
    "ptx_op_rows_scatter": (_I, [_P, _P, _L, _I, _P, _P]),
    "ptx_op_out_positions": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "ptx_op_affine_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ptx_op_affine_bwd_list": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ptx_op_tokens_finish": (_I, [_P, _P, _I, _I, _I, _P]),
    "ptx_op_tokens_finish_bwd": (_I, [_P, _I, _I, _I, _P]),
    "ptx_train_block_sizes": (_I, [C.POINTER(PtxTrainBlock), C.POINTER(_Z), C.POINTER(_Z), C.POINTER(_Z)]),
    "ptx_train_block_fwd": (_I, [C.POINTER(PtxTrainBlock), _P]),
    "ptx_train_block_bwd": (_I, [C.POINTER(PtxTrainBlock), _P]),
    "ptx_train_imgpool_sizes": (_I, [C.POINTER(PtxTrainImgPool), C.POINTER(_Z), C.POINTER(_Z), C.POINTER(_Z)]),
    "ptx_train_imgpool_fwd": (_I, [C.POINTER(PtxTrainImgPool), _P]),
    "ptx_train_imgpool_bwd": (_I, [C.POINTER(PtxTrainImgPool), _P]),
    "ptx_train_step_layout": (_I, [C.POINTER(PtxTrainStep), C.POINTER(PtxTrainStepLayout)]),
    "ptx_train_step_fwd": (_I, [C.POINTER(PtxTrainStep), _P]),
    "ptx_train_step_bwd": (_I, [C.POINTER(PtxTrainStep), _P]),
    "ptx_op_head_bwd_tmp_floats": (_Z, [_I, _I, _I]),
    "ptx_op_head_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "ptx_train_attn_tmp_floats": (_Z, [_I, _I, _I, _I, _I]),
    "ptx_train_attn_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _U64, _P, _P, _P, _P, _P]),
    "ptx_train_attn_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _U64, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
})

_lib: Optional[C.CDLL] = None


def use_test_hooks_library() -> None:
    """TEST ONLY: make this process load libproxyt_hip_testhooks.so (same C ABI + PTX_GATE_FAULT / PTX_GATE_TRAP).  Must be called
    before the first ``lib()``; raises once the product library is loaded."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_test_hooks_library(): a library is already loaded in this process")
    LIB_PATH = os.path.join(_HERE, "libproxyt_hip_testhooks.so")


def lib() -> C.CDLL:
    """Load (once) and type the shared library; fail loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is the only implementation of this "
            "path (no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C proxytransformation_amd/csrc`.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    got = handle.ptx_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libproxyt_hip.so ABI {got} != binding ABI {ABI_VERSION}")
    _lib = handle
    return handle


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ptx_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
