"""ctypes binding of libdruggen_hip.so -- the C ABI in include/druggen_hip.h.

This is the stub a reference maintainer would add: raw device
pointers from ``tensor.data_ptr()``, the caller's HIP stream, integer status
codes turned into ``RuntimeError(dg_last_error_string())``.  There is no CPU
fallback: if the shared object is missing or the tensors are not on a GPU the
call fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DG_LIB") or os.path.join(_PKG, "lib", "libdruggen_hip.so")   # DG_LIB: developer A/B builds

_P = c_void_p


class FFNFwdArgs(ctypes.Structure):
    """dg_ffn_fwd_args of include/druggen_hip.h."""
    _fields_ = [(n, _P) for n in ("x", "w1_packed", "b1", "w2_packed", "b2", "gamma", "beta", "y", "h", "relu_bits", "pre_ln",
                                  "mean", "rstd")] + [("R", c_int64), ("eps", c_float)]


class FFNBwdArgs(ctypes.Structure):
    """dg_ffn_bwd_args of include/druggen_hip.h."""
    _fields_ = [(n, _P) for n in ("x", "h", "relu_bits", "pre_ln", "mean", "rstd", "gamma", "w1_dgrad_packed", "w2_dgrad_packed",
                                  "dy", "dz_add", "dz", "dh", "dx", "dgamma", "dbeta", "dw1", "db1", "dw2", "db2", "workspace")] + [
        ("workspace_bytes", c_size_t), ("R", c_int64)]


# name -> (restype, argtypes); mirrors include/druggen_hip.h one to one
SIGNATURES = {
    "dg_version": (c_int, []),
    "dg_last_error_string": (c_char_p, []),
    "dg_hidden_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "dg_hidden_scale_offset": (c_size_t, [c_int64, c_int]),
    "dg_attn_core_fwd": (c_int, [_P] * 6 + [c_int, c_int, c_int, c_float, c_int, _P]),
    "dg_attn_core_bwd": (c_int, [_P] * 10 + [c_int, c_int, c_int, c_float, c_int, _P]),
    "dg_attn_core_bwd_add": (c_int, [_P] * 11 + [c_int, c_int, c_int, c_float, c_int, _P]),
    "dg_attn_core_bwd2": (c_int, [_P] * 16 + [c_int, c_int, c_int, c_float, c_int, _P]),
    "dg_attn_half_packed_bytes": (c_size_t, [c_int]),
    "dg_attn_half_pack": (c_int, [_P, _P, _P, c_int, _P]),
    "dg_attn_half_fwd": (c_int, [_P] * 14 + [c_int, c_int, c_int, c_float, c_float, c_int, _P]),
    "dg_attn_half_f32_fwd": (c_int, [_P] * 17 + [c_int, c_int, c_int, c_float, c_float, _P]),
    "dg_attn_half_f32_bwd1_workspace_bytes": (c_size_t, [c_int]),
    "dg_attn_half_f32_bwd1": (c_int, [_P] * 19 + [_P, c_size_t, c_int, c_int, c_int, c_float, _P]),
    "dg_attn_half_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dg_attn_half_bwd": (c_int, [_P] * 16 + [_P, c_size_t, c_int, c_int, c_int, c_float, c_int, _P]),
    "dg_ln_residual_fwd": (c_int, [_P] * 7 + [c_int64, c_int, c_float, c_int, _P]),
    "dg_ln_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "dg_ln_residual_bwd": (c_int, [_P] * 9 + [_P, c_size_t, c_int64, c_int, c_int, _P]),
    "dg_ln_residual_bwd_add": (c_int, [_P] * 10 + [_P, c_size_t, c_int64, c_int, c_int, _P]),
    "dg_ln_residual_bwd2": (c_int, [_P] * 10 + [_P, c_size_t, c_int64, c_int, c_int, _P]),
    "dg_linear_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "dg_linear_wgrad": (c_int, [_P] * 5 + [_P, c_size_t, c_int64, c_int, c_int, c_int, _P]),
    "dg_row_gemm_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dg_row_gemm_pack": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "dg_row_gemm_mask_words": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "dg_row_gemm": (c_int, [_P] * 3 + [c_int64, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_int, _P]),
    "dg_linear_wgrad3": (c_int, [_P] * 6 + [_P, c_size_t, c_int64, c_int, _P]),
    "dg_row_gemm_pack3": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "dg_row_gemm_lin3": (c_int, [_P] * 5 + [c_int64, _P, _P, _P, c_int, _P]),
    "dg_row_gemm_sum3": (c_int, [_P] * 5 + [c_int64, _P, c_int, _P]),
    "dg_linear_wgrad_batch_begin": (c_int, []),
    "dg_linear_wgrad_batch_end": (c_int, [_P]),
    "dg_launch_pair_begin": (c_int, []),
    "dg_launch_pair_end": (c_int, [_P]),
    "dg_skinny_linear_fwd": (c_int, [_P] * 4 + [c_int64, c_int, c_int, c_int, _P]),
    "dg_skinny_linear_dgrad": (c_int, [_P] * 3 + [c_int64, c_int, c_int, c_int, _P]),
    "dg_skinny_linear_wgrad": (c_int, [_P] * 5 + [c_size_t, c_int64, c_int, c_int, c_int, _P]),
    "dg_embed_node_chain": (c_int, [_P] * 9 + [c_int64, c_int, c_int, _P]),
    "dg_embed_node_bwd": (c_int, [_P] * 8 + [c_int64, c_int, c_int, _P]),
    "dg_head_chain": (c_int, [_P] * 14 + [c_int64, c_int, _P]),
    "dg_head_bwd": (c_int, [_P] * 10 + [c_int64, c_int, _P]),
    "dg_head_wgrad": (c_int, [_P] * 12 + [c_int64, _P]),
    "dg_row_gemm_pack_batch": (c_int, [_P, c_int, c_int, c_int, _P]),
    "dg_row_gemm_ln_bwd_workspace_bytes": (c_size_t, [c_int]),
    "dg_row_gemm_ln_bwd": (c_int, [_P] * 3 + [c_int64, c_int] + [_P] * 7 + [_P, c_size_t, c_int, _P]),
    "dg_row_gemm_ln_bwd_in": (c_int, [_P] * 10 + [_P, c_size_t, c_int64, c_int, c_int, c_int, _P]),
    "dg_edge_ffn_ln_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "dg_edge_ffn_ln_fwd": (c_int, [_P] * 13 + [c_int64, c_int, c_int, c_float, c_int, _P]),
    "dg_edge_ffn_ln_bwd": (c_int, [_P] * 20 + [_P, c_size_t, c_int64, c_int, c_int, c_int, _P]),
    "dg_edge_ffn_ln_fwd_pair": (c_int, [ctypes.POINTER(FFNFwdArgs), ctypes.POINTER(FFNFwdArgs), c_int, c_int, c_int, _P]),
    "dg_edge_ffn_ln_bwd_pair": (c_int, [ctypes.POINTER(FFNBwdArgs), ctypes.POINTER(FFNBwdArgs), c_int, c_int, c_int, _P]),
    "dg_ffn_f32_packed_bytes": (c_size_t, []),
    "dg_ffn_f32_pack": (c_int, [_P, _P, _P, _P]),
    "dg_ffn_ln_fwd_f32": (c_int, [ctypes.POINTER(FFNFwdArgs), ctypes.POINTER(FFNFwdArgs), _P]),
    "dg_ffn_bf16_padded_rows": (c_int64, [c_int64]),
    "dg_ffn_bf16_packed_bytes": (c_size_t, []),
    "dg_ffn_bf16_pack": (c_int, [_P, _P, _P, _P]),
    "dg_ffn_bf16_mask_words": (c_size_t, [c_int64]),
    "dg_ffn_bf16_workspace_bytes": (c_size_t, [c_int64]),
    "dg_ffn_ln_fwd_bf16": (c_int, [_P] * 11 + [c_int64, c_float, _P]),
    "dg_ffn_ln_bwd_bf16": (c_int, [_P] * 18 + [_P, c_size_t, c_int64, _P]),
    "dg_embed_sym_packed_floats": (c_size_t, []),
    "dg_embed_sym_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dg_embed_sym_dgrad_packed_floats": (c_size_t, []),
    "dg_embed_sym_pack_dgrad": (c_int, [_P, _P, _P]),
    "dg_embed_sym_pack": (c_int, [_P, _P, _P]),
    "dg_embed_sym_fwd": (c_int, [_P] * 6 + [c_int] * 7 + [_P]),
    "dg_embed_sym_bwd": (c_int, [_P] * 12 + [_P, c_size_t] + [c_int] * 7 + [_P]),
    "dg_embed_sym_bwd_bf16_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dg_embed_sym_bwd_bf16": (c_int, [_P] * 11 + [_P, c_size_t] + [c_int] * 6 + [_P]),
    "dg_embed_sym_bwd2": (c_int, [_P] * 11 + [_P, c_size_t] + [c_int] * 7 + [_P]),
    "dg_onehot_embed_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dg_onehot_embed_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dg_onehot_embed_bwd": (c_int, [_P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, _P]),
    "dg_densify": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P]),
    "dg_adamw_flat": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int64, _P]),
    "dg_adamw_flat_devstep": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, _P, _P]),
    "dg_argmax_decode": (c_int, [_P, c_int64, c_int, _P, _P]),
    "dg_prof_enable": (c_int, [c_int]),
    "dg_prof_reset": (c_int, []),
    "dg_prof_read": (c_int, [c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_double)]),
    "dg_set_edge_rows": (c_int, [c_int64]),
    "dg_edge_rows": (c_int64, []),
}

KERNEL_IDS = {"attn_fwd": 0, "attn_bwd": 1, "attn_bwd2": 2, "ln_fwd": 3, "ln_bwd": 4, "ln_bwd2": 5, "linear_wgrad": 6, "row_gemm": 7, "embed_sym": 8, "ffn": 9, "ffn_wgrad": 10,
              "attn_half_fwd": 11, "attn_half_bwd": 12,
              "row_gemm_e128": 13, "row_gemm_e_n384": 14, "row_gemm_e_k384": 15,
              "linear_wgrad_e128": 16, "linear_wgrad_e_n384": 17, "linear_wgrad_e_k384": 18,
              "ffn_node": 19, "ffn_wgrad_node": 20, "ffn_f32": 21, "ffn_f32_node": 22}
EDGE_ROWS = 65536        # DG_EDGE_ROWS of include/druggen_hip.h: the default of dg_set_edge_rows()
_edge_rows = EDGE_ROWS


def edge_rows() -> int:
    """Row count from which a launch counts as edge-level (profiler / traffic keys, traversal direction)."""
    return _edge_rows


def set_edge_rows(rows: int = 0) -> None:
    """dg_set_edge_rows: process-wide; ``rows <= 0`` restores DG_EDGE_ROWS.  ``GANStep.step`` sets it to B N^2 / 2 (never
    below the default), between the node-level (B N, 2 B N) and edge-level (B N^2) row counts of its batch."""
    global _edge_rows
    rows = int(rows) if rows and rows > 0 else EDGE_ROWS
    if rows != _edge_rows:
        check(load().dg_set_edge_rows(rows), "dg_set_edge_rows")
        _edge_rows = rows

_lock = threading.Lock()
_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                f"{LIB_PATH} not found: build it with `python -m druggen_amd.build` "
                "(druggen_amd has no CPU / eager fallback)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
        return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().dg_last_error_string()
        raise RuntimeError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")


DTYPES = {torch.float32: 0, torch.bfloat16: 1}     # DG_DTYPE_F32 / DG_DTYPE_BF16 of include/druggen_hip.h
F32_H16 = 2      # DG_DTYPE_F32_H16: float32 activations, the [R,384] feed-forward hidden tensors as one scaled fp16 plane
F32_H24 = 3      # DG_DTYPE_F32_H24: ... as the top 24 bits of every float32 (three bytes per element)
F32_DH16, F32_DH24 = 4, 5      # dg_edge_ffn_ln_bwd(_pair) only: h float32, dh stored as F32_H16 / F32_H24
F32_H32 = 6      # DG_DTYPE_F32_H32: the forward's h as pre-split hi + lo fp16 planes under one row scale (float32 class)
F32_H32_DH16 = 7      # dg_edge_ffn_ln_bwd(_pair) only: h as F32_H32, dh as F32_H16
HIDDEN_CODES = (F32_H16, F32_H24, F32_H32)


def dt(t) -> int:
    """ABI dtype code of an activation tensor."""
    try:
        return DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError(f"druggen_amd kernels store activations as float32 or bfloat16; got {t.dtype}") from None


def ptr(t):
    """Device pointer of a contiguous float32 / bfloat16 GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("druggen_amd kernels need GPU tensors (no CPU fallback); got device "
                           f"{t.device}")
    if t.dtype not in DTYPES:
        raise RuntimeError(f"druggen_amd kernels are float32 / bfloat16; got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("druggen_amd kernels need contiguous tensors")
    return t.data_ptr()


def fptr(t):
    """Like ptr(), for arguments the ABI declares `float*` (parameters, statistics, weight gradients)."""
    if t is not None and t.dtype != torch.float32:
        raise RuntimeError(f"expected a float32 tensor, got {t.dtype}")
    return ptr(t)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t) -> int:
    """The caller's current HIP stream on t's device as a raw handle (asked for at every launch: a step issues ~600 of them,
    so the raw-handle query is used where this torch has it -- no Stream object per call)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


# ---- profiler ---------------------------------------------------------------
def prof_enable(on=True, kernels=None) -> None:
    """Enable HIP-event timing for all kernels (``on=True``), none (``False``) or the named ones."""
    if kernels is not None:
        mask = 0
        for k in kernels:
            mask |= 1 << KERNEL_IDS[k]
        if mask & (1 << 31):
            mask -= 1 << 32
    else:
        mask = -1 if on else 0
    check(load().dg_prof_enable(mask), "dg_prof_enable")


def prof_reset() -> None:
    check(load().dg_prof_reset(), "dg_prof_reset")


def prof_read(kernel: str):
    n, ms = c_int64(0), c_double(0.0)
    check(load().dg_prof_read(KERNEL_IDS[kernel], ctypes.byref(n), ctypes.byref(ms)), "dg_prof_read")
    return int(n.value), float(ms.value)
