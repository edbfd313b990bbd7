"""Twice-differentiable autograd wrappers around the HIP kernels.

Each op is a pair of ``torch.autograd.Function``s: the forward op, and its
backward expressed as a second Function whose own backward calls the
second-order kernel.  That keeps the reference's gradient penalty
(``src/model/loss.py:32-39``: ``autograd.grad(..., create_graph=True)`` followed
by ``d_loss.backward()``, ``train.py:367``) working unchanged on these modules.
"""
from __future__ import annotations

import contextlib
import ctypes
import threading
import weakref

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from .options import options

__all__ = ["attach_one_hot_labels", "attn_core", "ln_residual", "linear", "linear_relu", "linear_ln", "ffn_ln", "attn_block", "embed_sym", "inputs_only_backward",
           "second_order_forward", "in_second_order_forward", "readout", "traffic_reset", "traffic_bytes", "traffic_flops", "traffic_floor_bytes",
           "set_activation_dtype", "activation_dtype", "hidden_storage", "hidden_forward_storage", "hidden_to_float", "activations", "as_one_hot", "one_hot_labels", "embed_sym_onehot", "OutSlot", "join_parts", "set_fused_ffn_f32"]

# Algorithmic HBM bytes per kernel, accumulated per launch so that
# bench.py can turn the HIP-event times of dg_prof_* into achieved GB/s.
_traffic = {}


def traffic_reset() -> None:
    _traffic.clear()


def traffic_bytes(kernel: str) -> int:
    return _traffic.get(kernel, 0)


def _account(kernel: str, nbytes: int, flops: int = 0, floor: int = -1) -> None:
    """``nbytes``: what the launch must move given what it is asked to produce (inputs + outputs + tensors it saves for a
    backward).  ``floor``: inputs + outputs only (nothing saved, a backward recomputes) -- defaults to
    ``nbytes`` for launches that save nothing."""
    _traffic[kernel] = _traffic.get(kernel, 0) + nbytes
    _traffic[kernel + ":floor"] = _traffic.get(kernel + ":floor", 0) + (nbytes if floor < 0 else floor)
    if flops:
        _traffic[kernel + ":flops"] = _traffic.get(kernel + ":flops", 0) + flops


def traffic_floor_bytes(kernel: str) -> int:
    return _traffic.get(kernel + ":floor", 0)


def _gemm_key(R: int, K: int, N: int) -> str:
    """Profiler / traffic key of a row GEMM launch: edge-level launches by shape (DG_K_ROW_GEMM_E_*), the rest together."""
    if R < _lib.edge_rows():
        return "row_gemm"
    return "row_gemm_e_k384" if K == 384 else ("row_gemm_e_n384" if N == 384 else "row_gemm_e128")


def _wgrad_key(R: int, N: int, K: int) -> str:
    """Profiler / traffic key of a weight-gradient launch (DG_K_LINEAR_WGRAD_E_*: edge-level launches by shape of dW)."""
    if R >= _lib.edge_rows():
        if (N, K) == (128, 128):
            return "linear_wgrad_e128"
        if (N, K) == (384, 128):
            return "linear_wgrad_e_n384"
        if (N, K) == (128, 384):
            return "linear_wgrad_e_k384"
    return "linear_wgrad"


def traffic_flops(kernel: str) -> int:
    return _traffic.get(kernel + ":flops", 0)


def _dev(t):
    """Run the launch with t's device current (nn.DataParallel replica threads)."""
    if torch.cuda.current_device() == t.device.index:
        return contextlib.nullcontext()
    return torch.cuda.device(t.device)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------
# storage of the [R,384] hidden tensors of the float32 feed-forward (DG_DTYPE_F32_H16, include/druggen_hip.h)
# --------------------------------------------------------------------------
_HIDDEN_MODES = ("f32", "dh16", "dh24", "f24", "f16")


def hidden_storage() -> str:
    """Storage of the [R,384] hidden tensors of the float32 feed-forward (``options.hidden``, DG_HIDDEN at import):
      "f32"   plain float32;
      "dh16"  (default) the BACKWARD's hidden tensors -- dh = (dz W2) * m and its second-order twin -- as ONE fp16 plane per row
              under an exact power-of-two row scale (DG_DTYPE_F32_H16); the forward's h = relu(fc1 x) stays float32 class: on chip
              in the fused forward (dg_ffn_ln_fwd_f32), pre-split hi + lo planes in the two-launch forward (DG_DTYPE_F32_H32);
      "dh24"  the backward's tensors as the top 24 bits of every float32 (DG_DTYPE_F32_H24), h float32;
      "f24" / "f16"   h as well.  Rounding h perturbs the forward pass, and a perturbed forward flips ReLU masks in the layers
              behind it: gradient errors of the order of the SQUARE ROOT of the perturbation on small batches (two-molecule
              goldens: 1.4e-3 - 4e-3 with f24's 2^-17, up to 1.5e-2 with f16's 2^-11) -- outside the 1e-3 parity bar, labelled
              modes like the bf16 configuration.  dh only travels through linear maps: its rounding stays a rounding."""
    return options.hidden


def set_hidden_storage(mode: str) -> None:
    """``options.hidden = mode`` (one of f32, dh16, dh24, f24, f16)."""
    options.hidden = mode


def hidden_forward_storage() -> str:
    """Storage of the FORWARD's h = relu(fc1 x) when it goes through HBM in the default mode: "split" -- the float32-class
    hi / lo fp16 split under one row scale, done once by the launch that writes h (DG_DTYPE_F32_H32: fc2's launch only moves the
    planes, the weight gradient dW2 = dz^T h reads the hi plane alone) -- or "f32" (every other mode)."""
    return "split" if hidden_storage() == "dh16" else "f32"


def _hidden_code(adt) -> int:
    """ABI dtype code of the FORWARD's hidden tensor h for activations of ``adt``."""
    if adt == torch.float32:
        mode = hidden_storage()
        if mode in ("f16", "f24"):
            return _lib.F32_H16 if mode == "f16" else _lib.F32_H24
        if hidden_forward_storage() == "split":
            return _lib.F32_H32
    return _lib.DTYPES[adt]


def _hidden_code_bwd(adt) -> int:
    """ABI dtype code of the BACKWARD's hidden tensors (dh, and (t W1^T) * m of the second order)."""
    if adt == torch.float32:
        mode = hidden_storage()
        if mode in ("f16", "dh16"):
            return _lib.F32_H16
        if mode in ("f24", "dh24"):
            return _lib.F32_H24
    return _lib.DTYPES[adt]


def _ffn_bwd_codes(h, adt, R: int, H: int):
    """(dtype code of the dg_edge_ffn_ln_bwd call, storage code of dh): h's storage is what the forward chose, dh's what
    ``_hidden_code_bwd`` says now -- equal, or (h float32, dh narrow) the DG_DTYPE_F32_DH16 / _DH24 pairs."""
    if _is_h16(h):
        code = _hidden_code_of(h, R, H)
        if code == _lib.F32_H32:      # (h pre-split: dh is the fp16 plane of the default mode)
            return _lib.F32_H32_DH16, _lib.F32_H16
        return code, code
    dh_code = _hidden_code_bwd(adt)
    if dh_code == _lib.F32_H16:
        return _lib.F32_DH16, dh_code
    if dh_code == _lib.F32_H24:
        return _lib.F32_DH24, dh_code
    return _lib.DTYPES[adt], _lib.DTYPES[adt]


def _hidden_empty(R: int, H: int, adt, code: int, device):
    """An uninitialised [R,H] hidden tensor: float32 / bfloat16 [R,H], or (DG_DTYPE_F32_H16) the opaque byte buffer
    [R][H] fp16 + [R] float32 inverse row scales that only the kernels read."""
    if code in _lib.HIDDEN_CODES:
        return torch.empty(int(_lib.load().dg_hidden_bytes(R, H, code)), dtype=torch.uint8, device=device)
    return torch.empty(R, H, dtype=adt, device=device)


def _hrow_bytes(code: int, es: int, H: int) -> int:
    """Bytes per row of a hidden tensor (traffic accounting)."""
    if code == _lib.F32_H32:
        return 4 * H + 4
    return 2 * H + 4 if code == _lib.F32_H16 else (3 * H if code == _lib.F32_H24 else es * H)


def _is_h16(t) -> bool:
    return t is not None and t.dtype == torch.uint8


def _hptr(t):
    """Device pointer of a hidden tensor (either storage)."""
    return t.data_ptr() if _is_h16(t) else _lib.ptr(t)


def _hidden_code_of(buf, R: int, H: int = 384) -> int:
    """The ABI dtype code of a hidden buffer made by ``_hidden_empty`` (its size tells the storage)."""
    if buf.numel() == R * H * 3:
        return _lib.F32_H24
    plane = (R * H * 2 + 255) // 256 * 256
    return _lib.F32_H32 if buf.numel() == 2 * plane + 4 * R else _lib.F32_H16


def hidden_to_float(buf, R: int, H: int = 384):
    """Decode a DG_DTYPE_F32_H16 / _H24 buffer into a float32 [R,H] tensor (tests, probes)."""
    if _hidden_code_of(buf, R, H) == _lib.F32_H24:
        b = buf.view(R * H, 3).to(torch.int32)
        bits = (b[:, 0] << 8) | (b[:, 1] << 16) | (b[:, 2] << 24)
        return bits.view(torch.float32).view(R, H)
    off = int(_lib.load().dg_hidden_scale_offset(R, H))
    half = buf[:R * H * 2].view(torch.float16).view(R, H).float()
    if _hidden_code_of(buf, R, H) == _lib.F32_H32:      # hi plane | lo plane | scales
        half = half + buf[off:off + R * H * 2].view(torch.float16).view(R, H).float()
        off *= 2
    scale = buf[off:off + 4 * R].view(torch.float32)
    return half * scale[:, None]


# --------------------------------------------------------------------------
# graph attention core  (reference src/model/layers.py:119-134)
# --------------------------------------------------------------------------
def _attn_shapes(q, e):
    B, N, C = q.shape
    if tuple(e.shape) != (B, N, N, C):
        raise RuntimeError(f"attn_core: edge tensor {tuple(e.shape)} does not match node tensor {tuple(q.shape)}")
    return B, N, C


class _AttnCore(Function):
    @staticmethod
    def forward(ctx, q, k, v, e, alpha, need_s):
        q, k, v, e = _c(q), _c(k), _c(v), _c(e)
        B, N, C = _attn_shapes(q, e)
        lib = _lib.load()
        s = torch.empty_like(e) if need_s else None
        o = torch.empty_like(q)
        with _dev(q):
            _lib.check(lib.dg_attn_core_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(s),
                                            _lib.ptr(o), B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)), "dg_attn_core_fwd")
        _account("attn_fwd", q.element_size() * B * ((2 if need_s else 1) * N * N * C + 4 * N * C))
        ctx.save_for_backward(q, k, v, e)
        ctx.alpha = alpha
        ctx.set_materialize_grads(False)
        if s is None:
            s = q.new_empty(0)
            ctx.mark_non_differentiable(s)
        return s, o

    @staticmethod
    def backward(ctx, ws, wo):
        q, k, v, e = ctx.saved_tensors
        if wo is None:
            wo = torch.zeros_like(q)
        if ws is not None and ws.numel() == 0:
            ws = None
        dq, dk, dv, de = _AttnCoreBwd.apply(q, k, v, e, ws, wo, ctx.alpha)
        return dq, dk, dv, de, None, None


class _AttnCoreBwd(Function):
    @staticmethod
    def forward(ctx, q, k, v, e, ws, wo, alpha):
        B, N, C = _attn_shapes(q, e)
        ws = None if ws is None else _c(ws)
        wo = _c(wo)
        lib = _lib.load()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        de = torch.empty_like(e)
        with _dev(q):
            _lib.check(lib.dg_attn_core_bwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(ws),
                                            _lib.ptr(wo), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(de),
                                            B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)), "dg_attn_core_bwd")
        _account("attn_bwd", q.element_size() * B * ((3 if ws is not None else 2) * N * N * C + 7 * N * C))
        ctx.save_for_backward(q, k, v, e, ws, wo)
        ctx.alpha = alpha
        return dq, dk, dv, de

    @staticmethod
    @once_differentiable
    def backward(ctx, tq, tk, tv, te):
        q, k, v, e, ws, wo = ctx.saved_tensors
        B, N, C = _attn_shapes(q, e)
        tq, tk, tv, te = _c(tq), _c(tk), _c(tv), _c(te)
        lib = _lib.load()
        gq, gk, gv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        ge = torch.empty_like(e)
        gws = torch.empty_like(e) if ws is not None else None
        gwo = torch.empty_like(q)
        with _dev(q):
            _lib.check(lib.dg_attn_core_bwd2(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(ws),
                                             _lib.ptr(wo), _lib.ptr(tq), _lib.ptr(tk), _lib.ptr(tv), _lib.ptr(te),
                                             _lib.ptr(gq), _lib.ptr(gk), _lib.ptr(gv), _lib.ptr(ge), _lib.ptr(gws),
                                             _lib.ptr(gwo), B, N, C, ctx.alpha, _lib.dt(q), _lib.stream_of(q)),
                       "dg_attn_core_bwd2")
        _account("attn_bwd2", q.element_size() * B * ((5 if ws is not None else 3) * N * N * C + 11 * N * C))
        return gq, gk, gv, ge, gws, gwo, None


def attn_core(q, k, v, e, alpha: float, need_s: bool = True):
    """(s, o) of the edge-modulated per-channel attention.

    s[b,i,j,c] = alpha q[b,i,c] k[b,j,c] (e^2+e)[b,i,j,c];  o = sum_j softmax_j(s) v_j.
    With ``need_s=False`` the [B,N,N,C] score tensor is not written (Discriminator's
    last block never reads it, reference models.py:202-207) and ``s`` is None.
    """
    s, o = _AttnCore.apply(q, k, v, e, float(alpha), bool(need_s))
    return (s if need_s else None), o


# --------------------------------------------------------------------------
# residual + LayerNorm  (reference src/model/layers.py:185-192)
# --------------------------------------------------------------------------
_ws_cache = {}
_cache_lock = threading.Lock()      # nn.DataParallel replica threads insert into / sweep the module-level caches concurrently


def _scratch(ref, need, tag="ln"):
    """Per (device, stream, thread, tag) scratch buffer owned by the caller side (PyTorch).  The thread is part of the key
    because nn.DataParallel replicas are threads: two of them on ONE device and stream (device_ids=[0, 0]) would
    otherwise interleave a kernel of one call with the reduction of another over the same workspace."""
    key = (ref.device, _lib.stream_of(ref), threading.get_ident(), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=ref.device)
        with _cache_lock:
            if len(_ws_cache) > 256:      # nn.DataParallel starts fresh replica threads per forward: drop dead threads' buffers
                alive = {t.ident for t in threading.enumerate()}
                for k in [k for k in list(_ws_cache) if k[2] not in alive]:
                    _ws_cache.pop(k, None)
            _ws_cache[key] = buf
    return buf


def _workspace(ref, R, C):
    need = int(_lib.load().dg_ln_workspace_bytes(R, C))
    return _scratch(ref, need, "ln"), need


class _LNResidual(Function):
    @staticmethod
    def forward(ctx, a, r, gamma, beta, eps):
        a = _c(a)
        r = None if r is None else _c(r)
        C = a.shape[-1]
        R = a.numel() // C
        lib = _lib.load()
        y = torch.empty_like(a)
        mean = torch.empty(R, dtype=torch.float32, device=a.device)
        rstd = torch.empty(R, dtype=torch.float32, device=a.device)
        with _dev(a):
            _lib.check(lib.dg_ln_residual_fwd(_lib.ptr(a), _lib.ptr(r), _lib.fptr(_c(gamma)), _lib.fptr(_c(beta)),
                                              _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), R, C, eps,
                                              _lib.dt(a), _lib.stream_of(a)), "dg_ln_residual_fwd")
        _account("ln_fwd", a.element_size() * R * C * (3 if r is not None else 2))
        # the penalty's forward (r is None: ln1): the input leaves as an alias output, so that the second-order adjoint of
        # the input comes back to THIS node and joins dz inside the backward kernel (dz_add) instead of an engine add
        ctx.alias = bool(r is None and ctx.needs_input_grad[0] and in_second_order_forward() and _alias_outputs_enabled())
        if ctx.alias:
            a = a.view_as(a)
        ctx.save_for_backward(a, r, gamma, mean, rstd)
        ctx.set_materialize_grads(False)
        return (y, a) if ctx.alias else y

    @staticmethod
    def backward(ctx, dy, ga=None):
        a, r, gamma, mean, rstd = ctx.saved_tensors
        want_aff = (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]) and not _inputs_only()
        if dy is None:
            dy = torch.zeros_like(a)
        dz, dgamma, dbeta = _LNResidualBwd.apply(a, r, gamma, mean, rstd, dy, want_aff, ga)
        return dz, (dz if r is not None else None), dgamma, dbeta, None


class _LNResidualBwd(Function):
    @staticmethod
    def forward(ctx, a, r, gamma, mean, rstd, dy, want_aff=True, dz_add=None):
        dy = _c(dy if dy.dtype == a.dtype else dy.to(a.dtype))
        if dz_add is not None:
            dz_add = _c(dz_add if dz_add.dtype == a.dtype else dz_add.to(a.dtype))
        C = a.shape[-1]
        R = a.numel() // C
        lib = _lib.load()
        dz = torch.empty_like(a)
        dgamma, dbeta = (torch.empty_like(gamma), torch.empty_like(gamma)) if want_aff else (None, None)
        with _dev(a):
            ws, need = _workspace(a, R, C)
            _lib.check(lib.dg_ln_residual_bwd_add(_lib.ptr(a), _lib.ptr(r), _lib.fptr(_c(gamma)), _lib.ptr(mean),
                                                  _lib.ptr(rstd), _lib.ptr(dy), _lib.ptr(dz_add), _lib.ptr(dz), _lib.ptr(dgamma),
                                                  _lib.ptr(dbeta), ws.data_ptr(), ws.numel(), R, C, _lib.dt(a),
                                                  _lib.stream_of(a)), "dg_ln_residual_bwd")
        _account("ln_bwd", a.element_size() * R * C * ((4 if r is not None else 3) + (dz_add is not None)))
        ctx.third = dz_add is not None
        ctx.save_for_backward(a, r, gamma, mean, rstd, dy)
        ctx.set_materialize_grads(False)
        return dz, dgamma, dbeta

    @staticmethod
    @once_differentiable
    def backward(ctx, tz, tgamma, tbeta):
        a, r, gamma, mean, rstd, dy = ctx.saved_tensors
        if tgamma is not None or tbeta is not None:
            # only reached when somebody differentiates parameter gradients again;
            # the WGAN-GP path differentiates the input gradient only (loss.py:32-39)
            raise RuntimeError("ln_residual: second-order terms through dgamma/dbeta are not implemented")
        if tz is None:
            return (None,) * 8
        if ctx.third:
            raise RuntimeError("ln_residual: third-order differentiation is not implemented")
        tz = _c(tz)
        C = a.shape[-1]
        R = a.numel() // C
        lib = _lib.load()
        gz, gdy = torch.empty_like(a), torch.empty_like(a)
        ggamma = torch.empty_like(gamma)
        with _dev(a):
            ws, need = _workspace(a, R, C)
            _lib.check(lib.dg_ln_residual_bwd2(_lib.ptr(a), _lib.ptr(r), _lib.fptr(_c(gamma)), _lib.ptr(mean),
                                               _lib.ptr(rstd), _lib.ptr(dy), _lib.ptr(tz), _lib.ptr(gz),
                                               _lib.ptr(gdy), _lib.ptr(ggamma), ws.data_ptr(), ws.numel(), R, C,
                                               _lib.dt(a), _lib.stream_of(a)), "dg_ln_residual_bwd2")
        _account("ln_bwd2", a.element_size() * R * C * (6 if r is not None else 5))
        return gz, (gz if r is not None else None), ggamma, None, None, gdy, None, None


def ln_residual(a, r, gamma, beta, eps: float = 1e-5):
    """LayerNorm(a + r) * gamma + beta over the last dim; ``r`` may be None."""
    out = _LNResidual.apply(a, r, gamma, beta, float(eps))
    return out[0] if isinstance(out, tuple) else out


# --------------------------------------------------------------------------
# nn.Linear with the weight/bias gradient on the fp32-MFMA split-K kernel
# (reference: every nn.Linear of src/model/layers.py; forward / input-gradient
# contractions stay on the ROCm BLAS behind F.linear / matmul)
# --------------------------------------------------------------------------
_ACT_DTYPES = {"f32": torch.float32, "fp32": torch.float32, "float32": torch.float32,
               "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


def _as_act_dtype(dtype):
    dtype = _ACT_DTYPES.get(dtype, dtype) if isinstance(dtype, str) else dtype
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(f"activation dtype must be torch.float32 or torch.bfloat16, got {dtype!r}")
    return dtype


def set_activation_dtype(dtype) -> None:
    """Storage type of the encoder activations produced by Generator / Discriminator from now on:
    ``torch.float32`` (BASELINE configs[1], default) or ``torch.bfloat16`` (configs[2]: every
    [B,N,N,C] / [B,N,C] activation and activation gradient lives in HBM as bf16, one bf16 MFMA per
    product, fp32 accumulation, fp32 softmax / LayerNorm statistics, fp32 parameters, optimizer
    state and weight gradients).  Model inputs and outputs (one-hot graphs, logits) stay float32."""
    _flags.act_dtype = _as_act_dtype(dtype)


def activation_dtype():
    return _flags.act_dtype


@contextlib.contextmanager
def activations(dtype):
    """``with activations(torch.bfloat16): ...`` -- scoped form of ``set_activation_dtype``."""
    prev = _flags.act_dtype
    _flags.act_dtype = _as_act_dtype(dtype)
    try:
        yield
    finally:
        _flags.act_dtype = prev


class _Flags:
    """Process-wide (NOT thread-local) pass flags.  A backward node of a CUDA tensor runs on the
    autograd engine's device thread, and nn.DataParallel runs every replica forward on its own thread:
    neither sees a ``threading.local`` set by the thread that entered the context manager (round-1
    bug: the gradient penalty's first-order pass still computed every weight gradient).  The contexts
    are entered by the CALLER of the model (``gradient_penalty``), never by a replica: the caller blocks
    until every replica thread / engine thread has finished, so all of them see one consistent value
    (``tests/test_hip_scale.py::test_dataparallel_replicas_on_one_device...`` runs two replica threads
    through the gradient penalty).  Two independent training loops in ONE process would race them."""
    inputs_only = 0
    second_order = 0
    act_dtype = torch.float32


_flags = _Flags()


def _inputs_only() -> bool:
    return _flags.inputs_only > 0


@contextlib.contextmanager
def inputs_only_backward():
    """Inside this context a backward pass skips parameter gradients of the
    custom ops.  Used around the gradient penalty's first-order
    ``autograd.grad(..., inputs=[int_node, int_edge])`` (loss.py:32-39), where
    PyTorch's built-in ops skip them too but custom Functions cannot tell."""
    _flags.inputs_only += 1
    try:
        yield
    finally:
        _flags.inputs_only -= 1


@contextlib.contextmanager
def _reduce_batch(ref, on=True):
    """dg_linear_wgrad_batch_begin / _end around a block's backward: the fixed-order reductions of its weight gradients
    (``_wgrad_many(..., open_batch=False)``) and of its LayerNorms' dgamma / dbeta (``_ln_bwd_rows(batch_slot=i)``) run
    as ONE launch at the end (at most 8 of them; further ones reduce at once)."""
    if not (on and ref.is_cuda):
        yield False
        return
    lib = _lib.load()
    with _dev(ref):
        _lib.check(lib.dg_linear_wgrad_batch_begin(), "dg_linear_wgrad_batch_begin")
        try:
            yield True
        finally:
            _lib.check(lib.dg_linear_wgrad_batch_end(_lib.stream_of(ref)), "dg_linear_wgrad_batch_end")


_pair_tls = threading.local()


def _pair_hold(*tensors) -> None:
    """Lifetime contract of riding launches (csrc/pair.h keeps RAW device pointers of a waiting launch until its carrier or
    dg_launch_pair_end): every operand of a launch issued inside ``_pair_launches`` is referenced from here until the
    region ends, so a temporary (a ``.contiguous()`` / ``.to()`` copy) cannot go back to the caching allocator -- and be
    handed to a kernel that is launched EARLIER in stream order -- while a rider still points at it."""
    keep = getattr(_pair_tls, "keep", None)
    if keep is not None:
        keep.extend(t for t in tensors if t is not None)


@contextlib.contextmanager
def _pair_launches(ref, on=True):
    """dg_launch_pair_begin / _end: node-level launches of the 384-wide row GEMMs and of the producer / consumer weight
    gradients issued inside wait for -- and ride in -- the next launch of the same kernel (include/druggen_hip.h).  The
    caller issues node, edge, node, edge ... and reads no waiting result before its carrier was called."""
    if not (on and ref.is_cuda):
        yield False
        return
    lib = _lib.load()
    outer = getattr(_pair_tls, "keep", None)
    if outer is None:
        _pair_tls.keep = []
    with _dev(ref):
        _lib.check(lib.dg_launch_pair_begin(), "dg_launch_pair_begin")
        try:
            yield True
        finally:
            try:      # (also on an exception path: whatever waits is launched before its operands can be freed)
                _lib.check(lib.dg_launch_pair_end(_lib.stream_of(ref)), "dg_launch_pair_end")
            finally:
                if outer is None:
                    _pair_tls.keep = None


def _wgrad_many(items, open_batch=True, pair_from=None):
    """[(dy2, x2, want_bias), ...] -> [(dW, db), ...]: the split-K kernels of up to 8 weight gradients run back to back
    into separate workspaces and ONE launch reduces them all (dg_linear_wgrad_batch_begin / _end) -- the six projections
    of an attention block used to cost six reduce launches.  Shapes outside the MFMA kernel take their usual path.
    ``pair_from``: items[pair_from:] are issued inside ``_pair_launches`` (a node-level item there rides in the next item
    of its shape; items before it launch at once -- a waiting launch nobody carries is slow, few workgroups)."""
    lib = _lib.load()
    ref = items[0][0]
    if any(isinstance(dy, tuple) for dy, _, _ in items):      # a (dq, dk, dv) triple: one stacked [384,128] gradient
        return _wgrad_many_mixed(items, open_batch, pair_from)
    dims = [_wgrad_dims(dy, x) for dy, x, _ in items]
    ok = (ref.is_cuda and len(items) <= 8
          and all((dy.dtype == x.dtype or _is_h16(dy) or _is_h16(x)) and d[1] > 16 and d[2] > 16 for (dy, x, _), d in zip(items, dims)))
    needs = [int(lib.dg_linear_wgrad_workspace_bytes(*d)) for d in dims] if ok else []
    if not ok or any(n == 0 for n in needs):
        if open_batch:
            return [_wgrad(dy, x, b) for dy, x, b in items]
        # the caller's batch is open: every reduce is deferred to its end, so the calls must not share the one "wgrad"
        # scratch buffer -- a private buffer per item
        out = []
        for i, (dy, x, b) in enumerate(items):
            n = int(lib.dg_linear_wgrad_workspace_bytes(*dims[i])) if dy.is_cuda else 0
            out.append(_wgrad(dy, x, b, ws=_scratch(dy, n, f"wgrad_fb{i}") if n else None))
        return out
    offs, total = [], 0
    for n in needs:
        offs.append(total)
        total += (n + 255) // 256 * 256
    out = []
    with _dev(ref):
        ws = _scratch(ref, total, "wgrad_batch")
        with _reduce_batch(ref, on=open_batch), contextlib.ExitStack() as pairing:
            for i, ((dy, x, b), off, n) in enumerate(zip(items, offs, needs)):
                if i == pair_from:
                    pairing.enter_context(_pair_launches(ref))
                out.append(_wgrad(dy, x, b, ws=ws[off:off + n]))
    return out


def _wgrad_many_mixed(items, open_batch=True, pair_from=None):
    """``_wgrad_many`` when an item's dy is a 3-tuple of [R,128] float32 matrices sharing x (``_wgrad3``).  Same batching:
    private workspaces, one reduce launch."""
    lib = _lib.load()
    ref = items[0][1]
    needs = [int(lib.dg_linear_wgrad_workspace_bytes(x.shape[0], 384 if isinstance(dy, tuple) else dy.shape[1], x.shape[1]))
             for dy, x, _ in items]
    offs, total = [], 0
    for n in needs:
        offs.append(total)
        total += (n + 255) // 256 * 256
    out = []
    with _dev(ref):
        ws = _scratch(ref, total, "wgrad_batch")
        with _reduce_batch(ref, on=open_batch), contextlib.ExitStack() as pairing:
            for i, ((dy, x, b), off, n) in enumerate(zip(items, offs, needs)):
                if i == pair_from:
                    pairing.enter_context(_pair_launches(ref))
                out.append(_wgrad3(dy, x, b, ws=ws[off:off + n]) if isinstance(dy, tuple) else _wgrad(dy, x, b, ws=ws[off:off + n]))
    return out


def _wgrad(dy2, x2, want_bias, dy_mask=None, ws=None):
    """dW [N,K] = dy2^T x2, db [N] = column sums of dy2 (or None); float32 results for float32 or
    bfloat16 operands.  ``ws``: a private workspace (calls inside ``_wgrad_many``)."""
    if _is_h16(dy2) or _is_h16(x2):      # a [R,384] hidden operand as fp16 plane + row scales (DG_DTYPE_F32_H16)
        return _wgrad_h16(dy2, x2, want_bias, ws)
    if dy2.dtype != x2.dtype:      # e.g. fp32 logit gradients against bf16 activations (readout layers)
        dy2 = dy2.to(x2.dtype)
    R, N = dy2.shape
    K = x2.shape[1]
    lib = _lib.load()
    need = int(lib.dg_linear_wgrad_workspace_bytes(R, N, K)) if dy2.is_cuda else 0
    if need == 0 and dy2.is_cuda and dy_mask is None and K <= 16 and int(lib.dg_linear_wgrad_workspace_bytes(R, K, N)):
        # few INPUT features (embedding layer 1, Linear(5 -> 64), reference models.py:57): the same
        # streaming kernel with the operands swapped gives dW^T
        dwt, _ = _wgrad(x2, dy2, False)
        return dwt.t().contiguous(), (dy2.float().sum(0) if want_bias else None)
    if need == 0:      # shape outside the kernel's table: library GEMM on the same device
        dyf, xf = dy2.float(), x2.float()
        if dy_mask is not None:
            dyf = dyf * (dy_mask > 0)
        return dyf.t().mm(xf), (dyf.sum(0) if want_bias else None)
    dw = torch.empty(N, K, dtype=torch.float32, device=dy2.device)
    db = torch.empty(N, dtype=torch.float32, device=dy2.device) if want_bias else None
    with _dev(dy2):
        if ws is None:
            ws = _scratch(dy2, need, "wgrad")
        _lib.check(lib.dg_linear_wgrad(_lib.ptr(dy2), _lib.ptr(dy_mask), _lib.ptr(x2), _lib.ptr(dw), _lib.ptr(db), ws.data_ptr(),
                                       ws.numel(), R, N, K, _lib.dt(dy2), _lib.stream_of(dy2)), "dg_linear_wgrad")
    _pair_hold(dy2, dy_mask, x2, dw, db, ws)
    _account(_wgrad_key(R, N, K), dy2.element_size() * R * (N * (2 if dy_mask is not None else 1) + K), 2 * R * N * K)
    return dw, db


def _wgrad_dims(dy2, x2):
    """(R, N, K) of a weight gradient whose 384-wide operand may be a DG_DTYPE_F32_H16 buffer."""
    if _is_h16(dy2):
        return x2.shape[0], 384, x2.shape[1]
    if _is_h16(x2):
        return dy2.shape[0], dy2.shape[1], 384
    return dy2.shape[0], dy2.shape[1], x2.shape[1]


def _wgrad_h16(dy2, x2, want_bias, ws=None):
    R, N, K = _wgrad_dims(dy2, x2)
    lib = _lib.load()
    other = x2 if _is_h16(dy2) else dy2
    if other.dtype != torch.float32 or (N, K) not in ((384, 128), (128, 384)):
        raise RuntimeError(f"weight gradient with an fp16 hidden operand: float32 [R,128] partner expected, got {other.dtype} N={N} K={K}")
    dw = torch.empty(N, K, dtype=torch.float32, device=other.device)
    db = torch.empty(N, dtype=torch.float32, device=other.device) if want_bias else None
    with _dev(other):
        if ws is None:
            ws = _scratch(other, int(lib.dg_linear_wgrad_workspace_bytes(R, N, K)), "wgrad")
        code = _hidden_code_of(dy2 if _is_h16(dy2) else x2, R)
        _lib.check(lib.dg_linear_wgrad(_hptr(dy2), None, _hptr(x2), _lib.ptr(dw), _lib.ptr(db), ws.data_ptr(), ws.numel(), R, N, K,
                                       code, _lib.stream_of(other)), "dg_linear_wgrad")
    _pair_hold(dy2, x2, dw, db, ws)
    _account(_wgrad_key(R, N, K), R * (4 * 128 + _hrow_bytes(_lib.F32_H16 if code == _lib.F32_H32 else code, 4, 384)), 2 * R * N * K)
    return dw, db


def _mm_rows(a, w, mode, bias=None):
    """a @ w^T (mode 0) or a @ w (mode 1) over the rows of ``a`` on dg_row_gemm when the shape is one
    of its three, else on the ROCm BLAS (tiny / odd layers: embedding, readout, discriminator head).
    The result has ``a``'s dtype (float32 or bfloat16 activations; parameters are float32)."""
    rows, cols = w.shape
    K, N = (cols, rows) if mode == 0 else (rows, cols)
    if a.is_cuda and a.dtype in _lib.DTYPES and row_gemm_supported(K, N):
        out = row_gemm(_c(a).reshape(-1, K), packed_weight(w, mode, a.dtype), K, N, bias=bias)
        return out.view(*a.shape[:-1], N)
    if a.dtype != w.dtype:
        w = w.to(a.dtype)
        bias = None if bias is None else bias.to(a.dtype)
    if mode == 0:
        return torch.nn.functional.linear(a, w, bias)
    return a.matmul(w)


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return _mm_rows(x, w, 0, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        need_w = ctx.needs_input_grad[1] and not _inputs_only()
        dx, dw, db = _LinearBwd.apply(x, w, dy, ctx.has_bias and need_w, ctx.needs_input_grad[0], need_w)
        return dx, dw, db


class _LinearBwd(Function):
    @staticmethod
    def forward(ctx, x, w, dy, want_bias, need_x, need_w):
        dy = _c(dy)
        N, K = w.shape
        dx = _mm_rows(dy, w, 1) if need_x else None
        dw = db = None
        if need_w:
            dw, db = _wgrad(dy.reshape(-1, N), _c(x).reshape(-1, K), want_bias)
        ctx.save_for_backward(x, w, dy)
        ctx.set_materialize_grads(False)
        return dx, dw, db

    @staticmethod
    @once_differentiable
    def backward(ctx, tdx, tdw, tdb):
        x, w, dy = ctx.saved_tensors
        N, K = w.shape
        g_x = g_w = g_dy = None
        if tdx is not None:
            tdx = _c(tdx)
            g_dy = _mm_rows(tdx, w, 0)
            if not _inputs_only():
                g_w, _ = _wgrad(dy.reshape(-1, N), tdx.reshape(-1, K), False)
        if tdw is not None:
            g_x = dy.matmul(tdw)
            t = x.matmul(tdw.t())
            g_dy = t if g_dy is None else g_dy + t
        if tdb is not None:
            g_dy = tdb.expand_as(dy) if g_dy is None else g_dy + tdb
        return g_x, g_w, g_dy, None, None, None


def linear(x, weight, bias=None):
    """``F.linear`` whose weight/bias gradients (first and second order) run on
    ``dg_linear_wgrad``."""
    return _Linear.apply(x, weight, bias)


class _Readout(Function):
    """nn.Linear(128 -> N <= 16) over edge / node rows with float32 logits whatever the activation dtype (reference
    models.py:67-68,100-101: readout_e / readout_n): one streaming kernel per direction (dg_skinny_linear_fwd / _dgrad,
    dg_skinny_linear_wgrad) instead of `x.float()` + a library GEMM.  First order; a graph that is differentiated again
    goes through the composite."""

    @staticmethod
    def forward(ctx, x, w, b):
        N, K = w.shape
        x2 = _c(x).reshape(-1, K)
        R = x2.shape[0]
        y = torch.empty(R, N, dtype=torch.float32, device=x.device)
        lib = _lib.load()
        with _dev(x2):
            _lib.check(lib.dg_skinny_linear_fwd(_lib.ptr(x2), _lib.fptr(_c(w)), _lib.fptr(None if b is None else _c(b)),
                                                _lib.ptr(y), R, N, K, _lib.dt(x2), _lib.stream_of(x2)),
                       "dg_skinny_linear_fwd")
        _account("readout", x2.element_size() * R * K + 4 * R * N)
        ctx.save_for_backward(x, w, b)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        if torch.is_grad_enabled():
            return _double_backward_fallback(lambda x_, w_, b_: torch.nn.functional.linear(x_.float(), w_, b_),
                                             (x, w, b), dy)
        N, K = w.shape
        lib = _lib.load()
        dy2 = _c(dy.float()).reshape(-1, N)
        x2 = _c(x).reshape(-1, K)
        R = x2.shape[0]
        dx = dw = db = None
        with _dev(x2):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x2)
                _lib.check(lib.dg_skinny_linear_dgrad(_lib.ptr(dy2), _lib.fptr(_c(w)), _lib.ptr(dx), R, N, K, _lib.dt(x2),
                                                      _lib.stream_of(x2)), "dg_skinny_linear_dgrad")
                _account("readout", x2.element_size() * R * K + 4 * R * N)
            if ctx.needs_input_grad[1] and not _inputs_only():
                dw = torch.empty_like(w)
                db = torch.empty(N, dtype=torch.float32, device=x.device) if b is not None else None
                ws = _scratch(x2, int(lib.dg_linear_wgrad_workspace_bytes(R, N, K)), "wgrad")
                _lib.check(lib.dg_skinny_linear_wgrad(_lib.ptr(dy2), _lib.ptr(x2), _lib.ptr(dw), _lib.ptr(db), ws.data_ptr(),
                                                      ws.numel(), R, N, K, _lib.dt(x2), _lib.stream_of(x2)),
                           "dg_skinny_linear_wgrad")
                _account("linear_wgrad", x2.element_size() * R * K + 4 * R * N)
        return (None if dx is None else dx.view(x.shape)), dw, db


_HEAD_ACTS = {"relu": 0, "leaky": 1}


class _NodeEmbed(Function):
    """Linear(E, 64) - act - Linear(64, 128) - act over the node rows (reference models.py:52-56, 154-158) as ONE launch
    (dg_embed_node_chain); backward: one launch for g2 / g1 / dz (dg_embed_node_bwd) + the two weight gradients on
    dg_linear_wgrad; differentiable again (``_NodeEmbedBwd``: the penalty's second order is the chain kernel with the
    activation pattern as a mask)."""

    @staticmethod
    def forward(ctx, z, w1, b1, w2, b2, act):
        E = z.shape[-1]
        z2 = _c(z).reshape(-1, E)
        R = z2.shape[0]
        a1 = torch.empty(R, 64, dtype=torch.float32, device=z.device)
        a2 = torch.empty(R, 128, dtype=torch.float32, device=z.device)
        lib = _lib.load()
        with _dev(z2):
            _lib.check(lib.dg_embed_node_chain(_lib.ptr(z2), None, None, _lib.fptr(_c(w1)), _lib.fptr(_c(b1)), _lib.fptr(_c(w2)),
                                               _lib.fptr(_c(b2)), _lib.ptr(a1), _lib.ptr(a2), R, E, act, _lib.stream_of(z2)),
                       "dg_embed_node_chain")
        ctx.save_for_backward(z2, a1, a2, w1, w2)
        ctx.act, ctx.zshape = act, z.shape
        return a2.view(*z.shape[:-1], 128)

    @staticmethod
    def backward(ctx, g):
        z2, a1, a2, w1, w2 = ctx.saved_tensors
        need_w = any(ctx.needs_input_grad[1:5]) and not _inputs_only()
        dz, dw1, db1, dw2, db2 = _NodeEmbedBwd.apply(g, z2, a1, a2, w1, w2, ctx.needs_input_grad[0], need_w, ctx.act)
        return (None if dz is None else dz.view(ctx.zshape)), dw1, db1, dw2, db2, None


class _NodeEmbedBwd(Function):
    @staticmethod
    def forward(ctx, g, z2, a1, a2, w1, w2, need_z, need_w, act):
        gshape = g.shape
        g = _c(g.float()).reshape(-1, 128)
        R, E = z2.shape
        dev = z2.device
        g2 = torch.empty(R, 128, dtype=torch.float32, device=dev)
        g1 = torch.empty(R, 64, dtype=torch.float32, device=dev)
        dz = torch.empty(R, E, dtype=torch.float32, device=dev) if need_z else None
        lib = _lib.load()
        with _dev(z2):
            _lib.check(lib.dg_embed_node_bwd(_lib.ptr(g), _lib.ptr(a1), _lib.ptr(a2), _lib.fptr(_c(w1)), _lib.fptr(_c(w2)),
                                             _lib.ptr(g2), _lib.ptr(g1), _lib.ptr(dz), R, E, act, _lib.stream_of(z2)),
                       "dg_embed_node_bwd")
        dw1 = db1 = dw2 = db2 = None
        if need_w:
            dw2, db2 = _wgrad(g2, a1, True)
            dw1, db1 = _wgrad(g1, z2, True)
        ctx.save_for_backward(z2, a1, a2, w1, w2, g1, g2)
        ctx.act, ctx.gshape = act, gshape
        ctx.set_materialize_grads(False)
        return dz, dw1, db1, dw2, db2

    @staticmethod
    @once_differentiable
    def backward(ctx, t_dz, *tw):
        if any(t is not None for t in tw):
            raise RuntimeError("node_embed: second-order terms through parameter gradients are not implemented")
        if t_dz is None:
            return (None,) * 9
        z2, a1, a2, w1, w2, g1, g2 = ctx.saved_tensors
        R, E = z2.shape
        t = _c(t_dz.float()).reshape(-1, E)
        dev = z2.device
        u1 = torch.empty(R, 64, dtype=torch.float32, device=dev)
        u2 = torch.empty(R, 128, dtype=torch.float32, device=dev)
        lib = _lib.load()
        with _dev(z2):
            _lib.check(lib.dg_embed_node_chain(_lib.ptr(t), _lib.ptr(a1), _lib.ptr(a2), _lib.fptr(_c(w1)), None, _lib.fptr(_c(w2)),
                                               None, _lib.ptr(u1), _lib.ptr(u2), R, E, ctx.act, _lib.stream_of(z2)),
                       "dg_embed_node_chain")
        gw1 = gw2 = None
        if not _inputs_only():
            gw1, _ = _wgrad(g1, t, False)
            gw2, _ = _wgrad(g2, u1, False)
        # act'' = 0: nothing reaches the forward's activations or z
        return u2.view(ctx.gshape), None, None, None, gw1, gw2, None, None, None


def node_embed_supported(z, l1, l2, act_name) -> bool:
    return (z.is_cuda and z.dtype == torch.float32 and act_name in _HEAD_ACTS and 1 <= z.shape[-1] <= 16
            and tuple(l1.weight.shape) == (64, z.shape[-1]) and tuple(l2.weight.shape) == (128, 64)
            and l1.bias is not None and l2.bias is not None and l1.weight.dtype == torch.float32)


def node_embed(z, l1, l2, act_name):
    """act(Linear(64, 128)(act(Linear(E, 64)(z)))) over the last dimension of ``z``: float32 [..., 128]."""
    return _NodeEmbed.apply(z, l1.weight, l1.bias, l2.weight, l2.bias, _HEAD_ACTS[act_name])


def _head_launch(fn, name, *args):
    _lib.check(fn(*args), name)


class _HeadTail(Function):
    """Tail of the Discriminator head after its first Linear (reference models.py:173-178, 207): act - Linear(64, 32) - act -
    Linear(32, 16) - act - Linear(16, 1) over the rows of ``z1`` as ONE launch (dg_head_chain); the backward is one launch
    for the input gradient (dg_head_bwd) and one for the six parameter gradients (dg_head_wgrad), itself differentiable
    (``_HeadTailBwd``: the gradient penalty's second order is the same chain kernel with the activation pattern as a mask)."""

    @staticmethod
    def forward(ctx, z1, w2, b2, w3, b3, w4, b4, act):
        z1 = _c(z1)
        R = z1.shape[0]
        dev = z1.device
        a1, a2, a3 = (torch.empty(R, n, dtype=torch.float32, device=dev) for n in (64, 32, 16))
        out = torch.empty(R, 1, dtype=torch.float32, device=dev)
        lib = _lib.load()
        with _dev(z1):
            _head_launch(lib.dg_head_chain, "dg_head_chain", _lib.ptr(z1), None, None, None, _lib.fptr(_c(w2)), _lib.fptr(_c(b2)),
                         _lib.fptr(_c(w3)), _lib.fptr(_c(b3)), _lib.fptr(_c(w4)), _lib.fptr(_c(b4)), _lib.ptr(a1), _lib.ptr(a2),
                         _lib.ptr(a3), _lib.ptr(out), R, act, _lib.stream_of(z1))
        ctx.save_for_backward(a1, a2, a3, w2, w3, w4)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, g_out):
        a1, a2, a3, w2, w3, w4 = ctx.saved_tensors
        need_w = any(ctx.needs_input_grad[1:7]) and not _inputs_only()
        g1, dw2, db2, dw3, db3, dw4, db4 = _HeadTailBwd.apply(g_out, a1, a2, a3, w2, w3, w4, need_w, ctx.act)
        return g1, dw2, db2, dw3, db3, dw4, db4, None


class _HeadTailBwd(Function):
    @staticmethod
    def forward(ctx, g_out, a1, a2, a3, w2, w3, w4, need_w, act):
        g_out = _c(g_out.float()).reshape(-1, 1)
        R = a1.shape[0]
        dev = a1.device
        g3, g2, g1 = (torch.empty(R, n, dtype=torch.float32, device=dev) for n in (16, 32, 64))
        lib = _lib.load()
        dws = [None] * 6
        with _dev(a1):
            st = _lib.stream_of(a1)
            _head_launch(lib.dg_head_bwd, "dg_head_bwd", _lib.ptr(g_out), _lib.ptr(a1), _lib.ptr(a2), _lib.ptr(a3), _lib.fptr(_c(w2)),
                         _lib.fptr(_c(w3)), _lib.fptr(_c(w4)), _lib.ptr(g3), _lib.ptr(g2), _lib.ptr(g1), R, act, st)
            if need_w:
                dw2, dw3, dw4 = torch.empty_like(w2), torch.empty_like(w3), torch.empty_like(w4)
                db2, db3, db4 = (torch.empty(n, dtype=torch.float32, device=dev) for n in (32, 16, 1))
                _head_launch(lib.dg_head_wgrad, "dg_head_wgrad", _lib.ptr(g_out), _lib.ptr(a3), _lib.ptr(g3), _lib.ptr(a2), _lib.ptr(g2),
                             _lib.ptr(a1), _lib.ptr(dw4), _lib.ptr(db4), _lib.ptr(dw3), _lib.ptr(db3), _lib.ptr(dw2), _lib.ptr(db2),
                             R, st)
                dws = [dw2, db2, dw3, db3, dw4, db4]
        ctx.save_for_backward(g_out, a1, a2, a3, w2, w3, w4, g2, g3)
        ctx.act = act
        ctx.set_materialize_grads(False)
        return (g1, *dws)

    @staticmethod
    @once_differentiable
    def backward(ctx, t1, *tw):
        if any(t is not None for t in tw):
            raise RuntimeError("head_tail: second-order terms through parameter gradients are not implemented")
        if t1 is None:
            return (None,) * 9
        g_out, a1, a2, a3, w2, w3, w4, g2, g3 = ctx.saved_tensors
        t1 = _c(t1.float())
        R = a1.shape[0]
        dev = a1.device
        u1, u2, u3 = (torch.empty(R, n, dtype=torch.float32, device=dev) for n in (64, 32, 16))
        uo = torch.empty(R, 1, dtype=torch.float32, device=dev)
        lib = _lib.load()
        gw2 = gw3 = gw4 = None
        with _dev(a1):
            st = _lib.stream_of(a1)
            _head_launch(lib.dg_head_chain, "dg_head_chain", _lib.ptr(t1), _lib.ptr(a1), _lib.ptr(a2), _lib.ptr(a3), _lib.fptr(_c(w2)),
                         None, _lib.fptr(_c(w3)), None, _lib.fptr(_c(w4)), None, _lib.ptr(u1), _lib.ptr(u2), _lib.ptr(u3),
                         _lib.ptr(uo), R, ctx.act, st)
            if not _inputs_only():
                gw2, gw3, gw4 = torch.empty_like(w2), torch.empty_like(w3), torch.empty_like(w4)
                _head_launch(lib.dg_head_wgrad, "dg_head_wgrad", _lib.ptr(g_out), _lib.ptr(u3), _lib.ptr(g3), _lib.ptr(u2), _lib.ptr(g2),
                             _lib.ptr(u1), _lib.ptr(gw4), None, _lib.ptr(gw3), None, _lib.ptr(gw2), None, R, st)
        # act'' = 0: nothing reaches the forward's activations
        return uo, None, None, None, gw2, gw3, gw4, None, None


def head_tail_supported(z1, layers, act_name) -> bool:
    """``layers`` = the three Linears after the head's first one."""
    return (z1.is_cuda and z1.dtype == torch.float32 and z1.dim() == 2 and act_name in _HEAD_ACTS
            and [tuple(l.weight.shape) for l in layers] == [(32, 64), (16, 32), (1, 16)]
            and all(l.bias is not None and l.weight.dtype == torch.float32 for l in layers))


def head_tail(z1, layers, act_name):
    """act(z1) -> Linear(64, 32) -> act -> Linear(32, 16) -> act -> Linear(16, 1): [R, 64] -> [R, 1]."""
    l2, l3, l4 = layers
    return _HeadTail.apply(z1, l2.weight, l2.bias, l3.weight, l3.bias, l4.weight, l4.bias, _HEAD_ACTS[act_name])


def readout(x, weight, bias=None):
    """float32 ``F.linear(x.float(), weight, bias)`` for the Generator's readouts (dim 128 -> edge / node classes)."""
    ok = (x.is_cuda and x.dtype in _lib.DTYPES and weight.dim() == 2 and weight.shape[1] == 128 and 1 <= weight.shape[0] <= 16
          and weight.dtype == torch.float32)
    if not ok:
        return linear(x.float(), weight, bias)
    return _Readout.apply(x, weight, bias)


# --------------------------------------------------------------------------
# fp32-MFMA row GEMM with fused prologue / epilogue (dg_row_gemm)
# --------------------------------------------------------------------------
_pack_cache = {}
_weights_epoch = 0


def bump_weights_epoch() -> None:
    """Invalidate every packed weight (both caches share this epoch).  Called by writers that change
    parameters behind autograd's back: ``FlatAdamW.step`` (raw kernel on the flat buffer) and
    ``GraphedGANStep`` (before capture, so that the first use after each optimizer step records its
    pack kernel into the graph, and after every replay, which updates weights without touching
    ``tensor._version``)."""
    global _weights_epoch
    _weights_epoch += 1
    if len(_pack_cache) + len(_embed_pack_cache) > 8192:
        with _cache_lock:
            for cache in (_pack_cache, _embed_pack_cache):
                for k in [k for k, v in list(cache.items()) if v[0]() is None]:
                    cache.pop(k, None)


_alias_canon = {}      # data pointer -> weakref of the parameter an alias output stands for


def _weight_alias(t):
    """A view of parameter ``t`` that a forward node returns as an extra output and hands to its differentiable backward
    node in place of ``t`` (second-order forward of the gradient penalty): the second-order gradient of the parameter
    then comes back to the forward node as the gradient of that output and joins the node's own parameter gradient in
    one multi-tensor add -- otherwise the autograd engine sums the two contributions of every parameter with a launch
    each (~50 tiny adds per step)."""
    a = t.view_as(t)
    with _cache_lock:
        if len(_alias_canon) > 4096:
            for k in [k for k, r in list(_alias_canon.items()) if r() is None]:
                _alias_canon.pop(k, None)
        _alias_canon[a.data_ptr()] = weakref.ref(t)
    return a


def _canon(w):
    """The parameter behind an alias made by ``_weight_alias`` (same storage, shape, version), else ``w``: the pack
    caches are keyed by the parameter object."""
    r = _alias_canon.get(w.data_ptr())
    o = r() if r is not None else None
    if (o is not None and o is not w and o.data_ptr() == w.data_ptr() and o.shape == w.shape and o.stride() == w.stride()
            and o._version == w._version and o.dtype == w.dtype):
        return o
    return w


def _join_alias_grads(own, extra):
    """own[i] += extra[i] where both exist (one multi-tensor launch), own[i] = extra[i] where only the latter does."""
    own = list(own)
    have, add = [], []
    for i, (o, g) in enumerate(zip(own, extra)):
        if g is None:
            continue
        if o is None:
            own[i] = g
        elif torch.is_grad_enabled():
            own[i] = o + g
        else:
            have.append(o)
            add.append(g if g.dtype == o.dtype else g.to(o.dtype))
    if have:
        torch._foreach_add_(have, add)
    return own


def _alias_outputs_enabled() -> bool:
    return options.penalty_wgrad == "joined"


def packed_weight(w, mode: int, dtype=torch.float32):
    """MFMA-fragment-ordered copy of an nn.Linear weight (mode 0: forward, 1: input
    gradient) for activations of ``dtype``, cached per (storage, version): re-packed only after an
    optimizer step."""
    w = _canon(w)
    key = (id(w), mode, dtype)
    hit = _pack_cache.get(key)
    if (hit is not None and hit[0]() is w and hit[1] == w._version and hit[3] == w.data_ptr()
            and hit[4] == _weights_epoch):
        return hit[2]
    if len(_pack_cache) > 4096:       # entries of dead tensors (e.g. DataParallel replicas)
        with _cache_lock:
            for k in [k for k, v in list(_pack_cache.items()) if v[0]() is None]:
                _pack_cache.pop(k, None)
    lib = _lib.load()
    rows, cols = w.shape
    n_out, k = (rows, cols) if mode == 0 else (cols, rows)
    code = _lib.DTYPES[dtype]
    packed = torch.empty(int(lib.dg_row_gemm_packed_bytes(n_out, k, code)), dtype=torch.uint8, device=w.device)
    wd = _c(w.detach())
    with _dev(w):
        _lib.check(lib.dg_row_gemm_pack(_lib.fptr(wd), packed.data_ptr(), rows, cols, mode, code, _lib.stream_of(w)),
                   "dg_row_gemm_pack")
    _pack_cache[key] = (weakref.ref(w), w._version, packed, w.data_ptr(), _weights_epoch)
    return packed


_pack3_cache = {}


def packed_weight3(w0, w1, w2, mode: int):
    """``packed_weight`` for the vertical stack [w0; w1; w2] of three float32 [128,128] weights (q / k / v of an attention
    block) as ONE operand: mode 0 -> the 128 -> 384 forward operand of ``lin3``, mode 1 -> the 384 -> 128 input-gradient
    operand of ``sum3`` (dg_row_gemm_pack3).  Cached per (storages, versions) like ``packed_weight``."""
    w0, w1, w2 = _canon(w0), _canon(w1), _canon(w2)
    key = (id(w0), id(w1), id(w2), mode)
    ws = (w0, w1, w2)
    hit = _pack3_cache.get(key)
    if (hit is not None and all(r() is w for r, w in zip(hit[0], ws)) and hit[1] == tuple(w._version for w in ws)
            and hit[3] == tuple(w.data_ptr() for w in ws) and hit[4] == _weights_epoch):
        return hit[2]
    if len(_pack3_cache) > 1024:
        with _cache_lock:
            for k in [k for k, v in list(_pack3_cache.items()) if any(r() is None for r in v[0])]:
                _pack3_cache.pop(k, None)
    lib = _lib.load()
    n_out, k = (384, 128) if mode == 0 else (128, 384)
    packed = torch.empty(int(lib.dg_row_gemm_packed_bytes(n_out, k, 0)), dtype=torch.uint8, device=w0.device)
    wd = [_c(w.detach()) for w in ws]
    with _dev(w0):
        _lib.check(lib.dg_row_gemm_pack3(_lib.fptr(wd[0]), _lib.fptr(wd[1]), _lib.fptr(wd[2]), packed.data_ptr(), 128, mode, 0,
                                         _lib.stream_of(w0)), "dg_row_gemm_pack3")
    _pack3_cache[key] = (tuple(weakref.ref(w) for w in ws), tuple(w._version for w in ws), packed,
                         tuple(w.data_ptr() for w in ws), _weights_epoch)
    return packed


def lin3_supported(x2, ws) -> bool:
    """Three Linear(128,128) per launch (dg_row_gemm_lin3 / _sum3, dg_linear_wgrad3): float32 rows on the fp16 hi + lo
    kernels.  DG_QKV=separate keeps three launches (A/B measurements)."""
    return (x2.is_cuda and x2.dtype == torch.float32 and x2.shape[-1] == 128
            and all(tuple(w.shape) == (128, 128) and w.dtype == torch.float32 for w in ws))


def lin3(x2, ws, bs):
    """(x2 w0^T + b0, x2 w1^T + b1, x2 w2^T + b2) in one launch; ``bs`` entries may be None."""
    R = x2.shape[0]
    lib = _lib.load()
    ys = [torch.empty(R, 128, dtype=x2.dtype, device=x2.device) for _ in range(3)]
    with _dev(x2):
        _lib.check(lib.dg_row_gemm_lin3(_lib.ptr(x2), packed_weight3(*ws, 0).data_ptr(), _lib.ptr(ys[0]), _lib.ptr(ys[1]),
                                        _lib.ptr(ys[2]), R, _lib.fptr(bs[0]), _lib.fptr(bs[1]), _lib.fptr(bs[2]), 0,
                                        _lib.stream_of(x2)), "dg_row_gemm_lin3")
    _account(_gemm_key(R, 128, 384), 4 * R * (128 + 384), 2 * R * 128 * 384)
    return ys


def sum3(a0, a1, a2, ws, residual=None):
    """a0 w0 + a1 w1 + a2 w2 (+ residual): the input gradient of three Linears that share their input, one launch."""
    R = a0.shape[0]
    lib = _lib.load()
    y = torch.empty(R, 128, dtype=a0.dtype, device=a0.device)
    with _dev(a0):
        _lib.check(lib.dg_row_gemm_sum3(_lib.ptr(a0), _lib.ptr(a1), _lib.ptr(a2), packed_weight3(*ws, 1).data_ptr(), _lib.ptr(y),
                                        R, _lib.ptr(residual), 0, _lib.stream_of(a0)), "dg_row_gemm_sum3")
    _account(_gemm_key(R, 384, 128), 4 * R * (384 + 128 * (1 + (residual is not None))), 2 * R * 384 * 128)
    return y


def _wgrad3(dys, x2, want_bias, ws=None):
    """dW [384,128] = [dy0 | dy1 | dy2]^T x2 (+ db [384]): three weight gradients in one launch (dg_linear_wgrad3)."""
    R = x2.shape[0]
    lib = _lib.load()
    dw = torch.empty(384, 128, dtype=torch.float32, device=x2.device)
    db = torch.empty(384, dtype=torch.float32, device=x2.device) if want_bias else None
    with _dev(x2):
        if ws is None:
            ws = _scratch(x2, int(lib.dg_linear_wgrad_workspace_bytes(R, 384, 128)), "wgrad")
        _lib.check(lib.dg_linear_wgrad3(_lib.ptr(dys[0]), _lib.ptr(dys[1]), _lib.ptr(dys[2]), _lib.ptr(x2), _lib.ptr(dw),
                                        _lib.ptr(db), ws.data_ptr(), ws.numel(), R, 0, _lib.stream_of(x2)), "dg_linear_wgrad3")
    _account(_wgrad_key(R, 384, 128), 4 * R * (384 + 128), 2 * R * 384 * 128)
    return dw, db


_repack_tables = {}


def repack_params(params) -> int:
    """Re-pack every cached float32 pack of ``params`` in ONE launch (dg_row_gemm_pack_batch) -- called by the
    optimizer right after it changed them, instead of ~110 single pack launches at their next uses.  Entries that are not
    refreshed here (other dtypes, stream capture, first use) take the lazy path in ``packed_weight``.  Returns the number
    of packs refreshed."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return 0      # the device table is built with a host -> device copy
    ids = {id(p) for p in params}
    total = 0
    for dtype in (torch.float32, torch.bfloat16):
        entries = []      # (cache, key, weights, packed, mode)
        for key, hit in _pack_cache.items():
            if key[0] in ids and key[2] == dtype:
                w = hit[0]()
                if (w is not None and w.is_cuda and w.is_contiguous() and hit[3] == w.data_ptr()
                        and (dtype == torch.float32 or (w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0))):
                    entries.append((_pack_cache, key, (w,), hit[2], key[1]))
        if dtype == torch.float32:
            for key, hit in _pack3_cache.items():      # stacks of three weights (q / k / v)
                if key[0] in ids:
                    ws = tuple(r() for r in hit[0])
                    if (all(w is not None and w.is_cuda and w.is_contiguous() for w in ws)
                            and hit[3] == tuple(w.data_ptr() for w in ws)):
                        entries.append((_pack3_cache, key, ws, hit[2], key[3]))
        if len(entries) >= 2:
            total += _repack_entries(entries, dtype)
    return total


def _repack_entries(entries, dtype) -> int:
    dev = entries[0][2][0].device
    entries = [e for e in entries if e[2][0].device == dev]
    sig = (dev, dtype, tuple((id(c), k) for c, k, _, _, _ in entries), tuple(p.data_ptr() for _, _, _, p, _ in entries))
    tab = _repack_tables.get(sig)
    if tab is None:
        if len(_repack_tables) > 16:
            _repack_tables.clear()
        rows = []
        for _, _, ws, packed, mode in entries:
            if len(ws) == 1:
                rows.append([ws[0].data_ptr(), packed.data_ptr(), ws[0].shape[0], ws[0].shape[1], mode, 0, 0])
            else:      # [w0; w1; w2]: 384 stacked rows
                rows.append([ws[0].data_ptr(), packed.data_ptr(), 384, ws[0].shape[1], mode, ws[1].data_ptr(), ws[2].data_ptr()])
        tab = torch.tensor(rows, dtype=torch.int64, device=dev)
        _repack_tables[sig] = tab
    lib = _lib.load()
    w0 = entries[0][2][0]
    with _dev(w0):
        _lib.check(lib.dg_row_gemm_pack_batch(tab.data_ptr(), len(entries),
                                              max(384 if len(ws) == 3 else max(ws[0].shape) for _, _, ws, _, _ in entries),
                                              _lib.DTYPES[dtype], _lib.stream_of(w0)), "dg_row_gemm_pack_batch")
    for cache, key, ws, packed, _ in entries:
        if len(ws) == 1:
            cache[key] = (weakref.ref(ws[0]), ws[0]._version, packed, ws[0].data_ptr(), _weights_epoch)
        else:
            cache[key] = (tuple(weakref.ref(w) for w in ws), tuple(w._version for w in ws), packed,
                          tuple(w.data_ptr() for w in ws), _weights_epoch)
    return len(entries)


def row_gemm_supported(K: int, N: int) -> bool:
    return (K == 128 and N in (128, 384)) or (K == 384 and N == 128)


def row_gemm(a2, packed, K, N, bias=None, relu=False, want_relu_bits=False, mask_bits=None, residual=None, ln=None,
             want_pre=False, R=None, code=None):
    """y = epi(a2 @ B): see include/druggen_hip.h.  ``ln=(gamma, beta, eps)`` selects the LayerNorm
    epilogue and returns (y, mean, rstd[, pre]); ``want_relu_bits`` additionally returns the packed
    ReLU mask (y, bits) that a later input-gradient launch of the same geometry takes as ``mask_bits``.
    ``a2`` (and ``residual``) may be float32 or bfloat16; y / pre have the same dtype.
    ``code`` = DG_DTYPE_F32_H16: a 384-wide operand is a hidden buffer (``_hidden_empty``) -- the result for N = 384, ``a2``
    for K = 384 (then ``R`` must be given: the buffer carries no shape)."""
    h16_in = _is_h16(a2)
    if h16_in:
        code = _hidden_code_of(a2, R, K)
    else:
        R = a2.shape[0]
        code = _lib.dt(a2) if code is None else code
    lib = _lib.load()
    ref = residual if h16_in and residual is not None else a2
    adt = torch.float32 if h16_in else a2.dtype
    es = 4 if h16_in else a2.element_size()
    dev = a2.device
    y = _hidden_empty(R, N, adt, code, dev) if N == 384 else torch.empty(R, N, dtype=adt, device=dev)
    mean = rstd = gamma = beta = pre = bits = None
    eps = 0.0
    if ln is not None and want_pre:
        pre = torch.empty(R, N, dtype=adt, device=dev)
    if ln is not None:
        gamma, beta, eps = ln
        mean = torch.empty(R, dtype=torch.float32, device=dev)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
    if want_relu_bits:
        bits = torch.empty(int(lib.dg_row_gemm_mask_words(R, K, N, code)), dtype=torch.int32, device=dev)
    if residual is not None and residual.dtype != adt:
        residual = residual.to(adt)
    with _dev(ref):
        _lib.check(lib.dg_row_gemm(_hptr(a2), packed.data_ptr(), _hptr(y), R, K, N, _lib.fptr(bias),
                                   1 if relu else 0, None if bits is None else bits.data_ptr(),
                                   None if mask_bits is None else mask_bits.data_ptr(), _lib.ptr(residual),
                                   _lib.fptr(gamma), _lib.fptr(beta), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(pre),
                                   float(eps), code, _lib.stream_of(ref)), "dg_row_gemm")
    _pair_hold(a2, packed, y, bias, bits, mask_bits, residual, gamma, beta, mean, rstd, pre)
    kb = _hrow_bytes(code, es, K) if K == 384 else es * K      # bytes per row of the A operand / of the result
    nb = _hrow_bytes(code, es, N) if N == 384 else es * N
    _account(_gemm_key(R, K, N), R * (kb + nb + es * N * ((residual is not None) + (pre is not None))), 2 * R * K * N,
             floor=R * (kb + nb + es * N * (residual is not None)))
    if ln is not None:
        return (y, mean, rstd, pre) if want_pre else (y, mean, rstd)
    return (y, bits) if want_relu_bits else y


# --------------------------------------------------------------------------
# fused layers built on dg_row_gemm (first-order fast path)
# --------------------------------------------------------------------------
@contextlib.contextmanager
def second_order_forward():
    """Forward passes run inside this context will be differentiated twice
    (gradient penalty, loss.py:28-39): modules then build their graph from the
    twice-differentiable ops (linear / ln_residual / attn_core) instead of the
    fused first-order ones, which would have to recompute."""
    _flags.second_order += 1
    try:
        yield
    finally:
        _flags.second_order -= 1


def in_second_order_forward() -> bool:
    return _flags.second_order > 0


def _double_backward_fallback(composite, inputs, grad_out):
    """Backward of a fused op when the caller asked for create_graph=True: rebuild the
    op from twice-differentiable pieces on the original (graph-attached) inputs."""
    with torch.enable_grad():
        # aliases: one input may be upstream of another (x feeds fc1 AND is the residual); the
        # gradient must stop at each input, the outer engine continues from there
        alias = [t.view_as(t) if isinstance(t, torch.Tensor) and t.requires_grad else t for t in inputs]
        out = composite(*alias)
        need = [t for t in alias if isinstance(t, torch.Tensor) and t.requires_grad]
        grads = iter(torch.autograd.grad(out, need, grad_out, create_graph=True, allow_unused=True))
    return tuple(next(grads) if (isinstance(t, torch.Tensor) and t.requires_grad) else None for t in alias)


def _fusable(x, w):
    N, K = w.shape
    return x.is_cuda and x.dtype in _lib.DTYPES and row_gemm_supported(K, N)


def _ln_bwd_rows(pre, gamma, mean, rstd, dy2, dz_add=None, want_affine=True, batch_slot=None):
    """LayerNorm backward over rows of the saved pre-LN sum -> (dz [+ dz_add], dgamma, dbeta).  ``want_affine`` False
    (input-gradient-only passes: loss.py:32-39, the D pass of the G step): no reduction launch for dgamma / dbeta.
    ``batch_slot`` (inside ``_reduce_batch``): the reduction joins the batch's single launch; the partial sums get a
    workspace of their own (slot index) because they must survive until the batch ends."""
    R, N = pre.shape
    lib = _lib.load()
    dz = torch.empty_like(pre)
    dgamma, dbeta = (torch.empty(2, gamma.numel(), dtype=gamma.dtype, device=pre.device).unbind(0) if want_affine
                     else (None, None))
    with _dev(pre):
        if batch_slot is None:
            ws, _ = _workspace(pre, R, N)
        else:
            ws = _scratch(pre, int(lib.dg_ln_workspace_bytes(R, N)), f"ln_batch{batch_slot}")
        _lib.check(lib.dg_ln_residual_bwd_add(_lib.ptr(pre), None, _lib.fptr(_c(gamma)), _lib.ptr(mean),
                                              _lib.ptr(rstd), _lib.ptr(dy2), _lib.ptr(dz_add), _lib.ptr(dz),
                                              _lib.ptr(dgamma), _lib.ptr(dbeta), ws.data_ptr(), ws.numel(), R, N,
                                              _lib.dt(pre), _lib.stream_of(pre)), "dg_ln_residual_bwd")
    _account("ln_bwd", pre.element_size() * R * N * (4 if dz_add is not None else 3))
    return dz, dgamma, dbeta


def row_gemm_ln_bwd_supported(a2, K: int) -> bool:
    """dg_row_gemm_ln_bwd serves float32 rows, K = N = 128 (options.ln_bwd_epilogue: the equivalence tests' hook)."""
    return a2.is_cuda and a2.dtype == torch.float32 and K == 128 and options.ln_bwd_epilogue


def row_gemm_ln_bwd(a2, packed, K, residual, pre, gamma, mean, rstd):
    """(dz, dgamma, dbeta) of a LayerNorm whose output gradient is ``a2 @ B + residual``: the input-gradient GEMM
    with the LayerNorm backward as its epilogue (dg_row_gemm_ln_bwd) -- the gradient itself never reaches HBM."""
    R = a2.shape[0]
    lib = _lib.load()
    dz = torch.empty(R, 128, dtype=a2.dtype, device=a2.device)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
    code = _lib.dt(a2)
    with _dev(a2):
        ws = _scratch(a2, int(lib.dg_row_gemm_ln_bwd_workspace_bytes(code)), "lnb")
        _lib.check(lib.dg_row_gemm_ln_bwd(_lib.ptr(a2), packed.data_ptr(), _lib.ptr(dz), R, K, _lib.ptr(residual),
                                          _lib.ptr(pre), _lib.ptr(mean), _lib.ptr(rstd), _lib.fptr(_c(gamma)),
                                          _lib.ptr(dgamma), _lib.ptr(dbeta), ws.data_ptr(), ws.numel(), code,
                                          _lib.stream_of(a2)), "dg_row_gemm_ln_bwd")
    _account(_gemm_key(R, K, 128), a2.element_size() * R * (K + 128 * (2 + (residual is not None))), 2 * R * K * 128)
    return dz, dgamma, dbeta


def ln_bwd_row_gemm_supported(a2, K: int, N: int) -> bool:
    """dg_row_gemm_ln_bwd_in serves float32 rows, K = N = 128 (options.ln_bwd_prologue: the equivalence tests' hook)."""
    return a2.is_cuda and a2.dtype == torch.float32 and K == 128 and N == 128 and options.ln_bwd_prologue


def ln_bwd_row_gemm(pre, gamma, mean, rstd, dy2, packed, want_affine=True, batch_slot=None):
    """(dz, y, dgamma, dbeta) with dz = LayerNormBackward(dy2) and y = dz @ B in ONE launch (dg_row_gemm_ln_bwd_in):
    the producer waves of the GEMM run the LayerNorm backward on the rows they stream, dz is written once and never
    read back by this GEMM.  ``batch_slot``: as in ``_ln_bwd_rows``."""
    R = pre.shape[0]
    lib = _lib.load()
    dz = torch.empty_like(pre)
    y = torch.empty(R, 128, dtype=pre.dtype, device=pre.device)
    dgamma, dbeta = (torch.empty(2, gamma.numel(), dtype=gamma.dtype, device=pre.device).unbind(0) if want_affine
                     else (None, None))
    code = _lib.dt(pre)
    with _dev(pre):
        ws = _scratch(pre, int(lib.dg_row_gemm_ln_bwd_workspace_bytes(code)),
                      "lna" if batch_slot is None else f"lna_batch{batch_slot}")
        _lib.check(lib.dg_row_gemm_ln_bwd_in(_lib.ptr(dy2), _lib.ptr(pre), _lib.ptr(mean), _lib.ptr(rstd),
                                             _lib.fptr(_c(gamma)), packed.data_ptr(), _lib.ptr(dz), _lib.ptr(y),
                                             _lib.ptr(dgamma), _lib.ptr(dbeta), ws.data_ptr(), ws.numel(), R, 128, 128,
                                             code, _lib.stream_of(pre)), "dg_row_gemm_ln_bwd_in")
    _account(_gemm_key(R, 128, 128), pre.element_size() * R * 128 * 4, 2 * R * 128 * 128)
    return dz, y, dgamma, dbeta


def _ln_bwd2_rows(pre, gamma, mean, rstd, dy2, tz):
    """Backward of ``_ln_bwd_rows`` w.r.t. the adjoint ``tz`` of dz -> (gz, gdy, ggamma)."""
    R, N = pre.shape
    lib = _lib.load()
    gz, gdy = torch.empty_like(pre), torch.empty_like(pre)
    ggamma = torch.empty_like(gamma)
    with _dev(pre):
        ws, _ = _workspace(pre, R, N)
        _lib.check(lib.dg_ln_residual_bwd2(_lib.ptr(pre), None, _lib.fptr(_c(gamma)), _lib.ptr(mean), _lib.ptr(rstd),
                                           _lib.ptr(dy2), _lib.ptr(tz), _lib.ptr(gz), _lib.ptr(gdy), _lib.ptr(ggamma),
                                           ws.data_ptr(), ws.numel(), R, N, _lib.dt(pre), _lib.stream_of(pre)),
                   "dg_ln_residual_bwd2")
    _account("ln_bwd2", pre.element_size() * R * N * 5)
    return gz, gdy, ggamma


def _fused_ffn_enabled() -> bool:
    """options.ffn_bf16 = "unfused" keeps the bf16 feed-forward on the two-launch row-GEMM path (A/B measurements)."""
    return options.ffn_bf16 == "fused"


def _composite_ffn_ln(x, w1, b1, w2, b2, gamma, beta, eps):
    return ln_residual(x, linear(torch.relu(linear(x, w1, b1)), w2, b2), gamma, beta, eps)


_ffn_f32_pack_cache = {}


def _ffn_packed_f32(w1, w2):
    """The fragment-order copy of (fc1.weight [384,128], fc2.weight [128,384]) that the fused float32 feed-forward forward
    streams (dg_ffn_f32_pack), cached like ``packed_weight``: re-packed after an optimizer step."""
    w1, w2 = _canon(w1), _canon(w2)
    key = (id(w1), id(w2))
    hit = _ffn_f32_pack_cache.get(key)
    if (hit is not None and hit[0]() is w1 and hit[1]() is w2 and hit[2] == (w1._version, w2._version)
            and hit[4] == (w1.data_ptr(), w2.data_ptr()) and hit[5] == _weights_epoch):
        return hit[3]
    if len(_ffn_f32_pack_cache) > 1024:
        with _cache_lock:
            for k in [k for k, v in list(_ffn_f32_pack_cache.items()) if v[0]() is None or v[1]() is None]:
                _ffn_f32_pack_cache.pop(k, None)
    lib = _lib.load()
    packed = torch.empty(int(lib.dg_ffn_f32_packed_bytes()), dtype=torch.uint8, device=w1.device)
    with _dev(w1):
        _lib.check(lib.dg_ffn_f32_pack(_lib.fptr(_c(w1.detach())), _lib.fptr(_c(w2.detach())), packed.data_ptr(),
                                       _lib.stream_of(w1)), "dg_ffn_f32_pack")
    _ffn_f32_pack_cache[key] = (weakref.ref(w1), weakref.ref(w2), (w1._version, w2._version), packed,
                                (w1.data_ptr(), w2.data_ptr()), _weights_epoch)
    return packed


def set_fused_ffn_f32(on: bool) -> None:
    """Route the float32 feed-forward FORWARD through the fused kernel (dg_ffn_ln_fwd_f32: the [R,384] hidden tensor stays on
    chip; default) or through the two row-GEMM launches (``options.ffn_f32``; DG_FFN_F32=unfused at import)."""
    options.ffn_f32 = "fused" if on else "unfused"


def fused_ffn_f32_supported(x2, w1, w2) -> bool:
    """dg_ffn_ln_fwd_f32 serves float32 rows, dim 128, hidden 384, in the default hidden-storage mode: what it leaves for the
    backward is the hi fp16 plane of h (a DG_DTYPE_F32_H16 buffer) -- exactly what the default mode's backward reads of the
    pre-split h (dW2 = dz^T h_hi)."""
    return (options.ffn_f32 == "fused" and x2.is_cuda and x2.dtype == torch.float32 and tuple(w1.shape) == (384, 128)
            and tuple(w2.shape) == (128, 384) and hidden_storage() == "dh16")


def _ffn_f32_fwd_args(p, keep):
    """dg_ffn_fwd_args of one problem for dg_ffn_ln_fwd_f32 (``p``: the dict built by the feed-forward nodes)."""
    return _lib.FFNFwdArgs(
        _lib.ptr(p["x2"]), _ffn_packed_f32(p["w1"], p["w2"]).data_ptr(), _lib.fptr(_c(p["b1"])), None, _lib.fptr(_c(p["b2"])),
        _lib.fptr(_c(p["gamma"])), _lib.fptr(_c(p["beta"])), _lib.ptr(p["y"]), _hptr(p["h"]) if keep else None,
        p["bits"].data_ptr() if keep else None, _lib.ptr(p["pre"]), _lib.ptr(p["mean"]), _lib.ptr(p["rstd"]), p["R"], float(p["eps"]))


def _account_ffn_f32(R, C, H, keep):
    """Traffic of one problem of a fused forward launch: x in, y out (+ pre-LN sum, the hi plane of h with its row scales, one
    mask bit per hidden element when a backward follows); the floor is x in + y out."""
    key = "ffn_f32" if R >= _lib.edge_rows() else "ffn_f32_node"
    _account(key, R * (4 * C * (3 if keep else 2) + ((2 * H + 4 + H // 8) if keep else 0)), 4 * R * C * H, floor=R * 4 * C * 2)


class _FFNLN(Function):
    """LN(x + fc2(relu(fc1 x))) -- MLP + residual + LayerNorm of Encoder_Block (reference
    layers.py:50-53,191-192).  Forward: two row-GEMM launches (bias+ReLU epilogue; bias +
    residual + LayerNorm epilogue).  Backward: LN backward, then the fc2 input gradient with the
    ReLU mask applied in its epilogue, the fc1 input gradient with the residual gradient added in
    its epilogue, and the two weight gradients on the split-K kernel."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps):
        H, C = w1.shape
        x2 = _c(x).reshape(-1, C)
        R = x2.shape[0]
        lib = _lib.load()
        dev = x2.device
        adt, code, es = x2.dtype, _hidden_code(x2.dtype), x2.element_size()
        # no input needs a gradient (e.g. the Generator's forward inside the D step): nothing is kept for a backward --
        # no pre-LayerNorm sum (one [R,C] write pass) and no ReLU bit mask
        keep = any(ctx.needs_input_grad)
        fused = fused_ffn_f32_supported(x2, w1, w2)
        if fused:      # h stays on chip; its hi fp16 plane leaves for the backward's dW2 (a DG_DTYPE_F32_H16 buffer)
            code = _lib.F32_H16
        y = torch.empty(R, C, dtype=adt, device=dev)
        h = _hidden_empty(R, H, adt, code, dev) if (keep or not fused) else None
        pre = torch.empty(R, C, dtype=adt, device=dev) if keep else None
        mean = torch.empty(R, dtype=torch.float32, device=dev)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        bits = torch.empty(int(lib.dg_row_gemm_mask_words(R, C, H, code)), dtype=torch.int32, device=dev) if keep else None
        with _dev(x2):
            if fused:
                arg = _ffn_f32_fwd_args(dict(x2=x2, w1=w1, w2=w2, b1=b1, b2=b2, gamma=gamma, beta=beta, y=y, h=h, bits=bits, pre=pre,
                                             mean=mean, rstd=rstd, R=R, eps=eps), keep)
                _lib.check(lib.dg_ffn_ln_fwd_f32(None, ctypes.byref(arg), _lib.stream_of(x2)), "dg_ffn_ln_fwd_f32")
            else:
                _lib.check(lib.dg_edge_ffn_ln_fwd(_lib.ptr(x2), packed_weight(w1, 0, adt).data_ptr(), _lib.fptr(_c(b1)),
                                                  packed_weight(w2, 0, adt).data_ptr(), _lib.fptr(_c(b2)), _lib.fptr(_c(gamma)),
                                                  _lib.fptr(_c(beta)), _lib.ptr(y), _hptr(h),
                                                  None if bits is None else bits.data_ptr(),
                                                  _lib.ptr(pre), _lib.ptr(mean), _lib.ptr(rstd), R, C, H, eps, code,
                                                  _lib.stream_of(x2)), "dg_edge_ffn_ln_fwd")
        if fused:
            _account_ffn_f32(R, C, H, keep)
        else:
            hb = _hrow_bytes(code, es, H)
            _account(_gemm_key(R, C, H), R * (es * C + hb), 2 * R * C * H)
            _account(_gemm_key(R, H, C), R * (hb + es * (3 if keep else 2) * C), 2 * R * C * H, floor=R * (hb + es * 2 * C))
        if not keep:
            ctx.mark_non_differentiable(mean, rstd)
            return y.view(x.shape), None, mean, rstd
        ctx.save_for_backward(x, w1, b1, w2, b2, gamma, beta, h, mean, rstd, pre, bits)
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(mean, rstd)
        # `pre` (the pre-LayerNorm sum) is a second output so that (1) the gradient penalty's second order can hand
        # its adjoint back to THIS node: it then joins the LayerNorm gradient inside one backward pass instead of
        # triggering a second walk through fc2 / fc1; (2) the consumer of y can run this LayerNorm's backward in
        # the epilogue of its own input-gradient GEMM (``LNHandle``) and return the result as the gradient of `pre`.
        return y.view(x.shape), pre, mean, rstd

    @staticmethod
    def backward(ctx, dy, dpre, _dmean=None, _drstd=None):
        x, w1, b1, w2, b2, gamma, beta, h, mean, rstd, pre, bits = ctx.saved_tensors
        want_w = ctx.needs_input_grad[1] and not _inputs_only()
        want_aff = (ctx.needs_input_grad[5] or ctx.needs_input_grad[6]) and not _inputs_only()
        if dy is None and (dpre is None or torch.is_grad_enabled()):
            dy = torch.zeros_like(pre)
        # dy None, dpre given, no graph recorded: the LayerNorm backward already happened in the consumer's GEMM
        dx, dw1, db1, dw2, db2, dgamma, dbeta = _FFNLNBwd.apply(x, w1, b1, w2, b2, gamma, h, mean, rstd, pre, bits,
                                                                 dy, dpre, ctx.needs_input_grad[0], want_w, want_aff)
        return dx, dw1, db1, dw2, db2, dgamma, dbeta, None


class _FFNLNBwd(Function):
    """Backward of ``_FFNLN`` as a differentiable node: its own backward (the gradient penalty's second
    order, reference loss.py:32-39 + train.py:367) is again a sequence of row-GEMM / LayerNorm /
    weight-gradient launches.  With u = dz = LN'(z; dy), m the ReLU mask, dx = u + ((u W2) * m) W1:
        adj u  = t + ((t W1^T) * m) W2^T          adj W1 += ((u W2)*m)^T t      adj W2 += u^T ((t W1^T)*m)
        (adj z, adj gamma, adj dy) = LN''(z; dy, adj u)
    and adj z then runs the first-order backward of z = x + fc2(relu(fc1 x)) (no LayerNorm)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, h, mean, rstd, pre, bits, dy, dz_add, want_x, want_w, want_aff=None):
        if want_aff is None:
            want_aff = want_w
        H, C = w1.shape
        R = pre.shape[0]
        lib = _lib.load()
        dev = pre.device
        adt, es = pre.dtype, pre.element_size()
        code, dh_code = _ffn_bwd_codes(h, adt, R, H)
        x2 = _c(x).reshape(-1, C)
        dh = _hidden_empty(R, H, adt, dh_code, dev)
        dx = torch.empty(R, C, dtype=adt, device=dev) if want_x else None
        if dy is None:      # dz_add IS the LayerNorm input gradient (made by dg_row_gemm_ln_bwd in the consumer of y)
            dy2 = dgamma = dbeta = None
            dz_add = dz = _c(dz_add if dz_add.dtype == adt else dz_add.to(adt)).reshape(-1, C)
        else:
            dy2 = _c(dy if dy.dtype == adt else dy.to(adt)).reshape(-1, C)
            dz = torch.empty(R, C, dtype=adt, device=dev)
            # adjacent in memory: their reduction then rides in the block's single reduce launch
            dgamma, dbeta = torch.empty(2, gamma.numel(), dtype=gamma.dtype, device=dev).unbind(0) if want_aff else (None, None)
        dw1 = db1 = dw2 = db2 = None
        if want_w:
            dw1 = torch.empty_like(w1)
            db1 = torch.empty(H, dtype=torch.float32, device=dev)
            dw2 = torch.empty_like(w2)
            db2 = torch.empty(C, dtype=torch.float32, device=dev)
        need = int(lib.dg_edge_ffn_ln_workspace_bytes(R, C, H))
        with _dev(pre):
            ws = _scratch(pre, need, "ffn")
            _lib.check(lib.dg_edge_ffn_ln_bwd(_lib.ptr(x2), _hptr(h), bits.data_ptr(), _lib.ptr(pre), _lib.ptr(mean),
                                              _lib.ptr(rstd), _lib.fptr(_c(gamma)), packed_weight(w1, 1, adt).data_ptr(),
                                              packed_weight(w2, 1, adt).data_ptr(), _lib.ptr(dy2),
                                              _lib.ptr(None if dz_add is None else _c(dz_add)), _lib.ptr(dz),
                                              _hptr(dh), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                              _lib.ptr(dw1), _lib.ptr(db1), _lib.ptr(dw2), _lib.ptr(db2),
                                              ws.data_ptr(), ws.numel(), R, C, H, code, _lib.stream_of(pre)),
                       "dg_edge_ffn_ln_bwd")
        if dy2 is not None:
            _account("ln_bwd", es * R * C * 3)
        hb, dhb = _hrow_bytes(_hidden_code_of(h, R, H) if _is_h16(h) else _lib.dt(pre), es, H), _hrow_bytes(dh_code, es, H)
        _account(_gemm_key(R, C, H), R * (es * C + dhb), 2 * R * C * H)
        if dx is not None:
            _account(_gemm_key(R, H, C), R * (dhb + es * 2 * C), 2 * R * C * H)
        if want_w:
            # (a pre-split h: the weight gradient reads its hi plane only)
            _account(_wgrad_key(R, C, H), R * (es * C + (2 * H + 4 if code == _lib.F32_H32_DH16 else hb)), 2 * R * C * H)
            _account(_wgrad_key(R, H, C), R * (es * C + dhb), 2 * R * C * H)
        ctx.save_for_backward(x, w1, w2, gamma, h, mean, rstd, pre, bits, dy2, dz, dh)
        ctx.had_add = dz_add is not None      # (includes the dy-None case: never differentiated again)
        ctx.set_materialize_grads(False)
        ctx.xshape = x.shape
        return (None if dx is None else dx.view(x.shape)), dw1, db1, dw2, db2, dgamma, dbeta

    @staticmethod
    @once_differentiable
    def backward(ctx, t_dx, t_dw1, t_db1, t_dw2, t_db2, t_dg, t_db):
        if any(t is not None for t in (t_dw1, t_db1, t_dw2, t_db2, t_dg, t_db)):
            raise RuntimeError("ffn_ln: second-order terms through parameter gradients are not implemented")
        if t_dx is None:
            return (None,) * 16
        if ctx.had_add:
            raise RuntimeError("ffn_ln: third-order differentiation is not implemented")
        x, w1, w2, gamma, h, mean, rstd, pre, bits, dy2, dz, dh = ctx.saved_tensors
        H, C = w1.shape
        adt = pre.dtype
        t = _c(t_dx if t_dx.dtype == adt else t_dx.to(adt)).reshape(-1, C)
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        code = _hidden_code_of(dh, t.shape[0], H) if _is_h16(dh) else _lib.dt(pre)
        vbar = row_gemm(t, pw(w1, 0), C, H, mask_bits=bits, code=code)   # (t W1^T) * m
        ubar = row_gemm(vbar, pw(w2, 0), H, C, residual=t, R=t.shape[0]) # t + vbar W2^T
        zbar, dybar, gbar = _ln_bwd2_rows(pre, gamma, mean, rstd, dy2, ubar)
        gw1 = gw2 = None
        if not _inputs_only():
            (gw1, _), (gw2, _) = _wgrad_many([(dh, t, False),              # ((u W2)*m)^T t
                                              (dz, vbar, False)])          # u^T ((t W1^T)*m)
        # dx depends on x only through the saved pre-LN sum z: its adjoint goes back to the forward
        # node (second output of _FFNLN), which runs ONE backward pass for both gradient sources.
        # inputs: x, w1, b1, w2, b2, gamma, h, mean, rstd, pre, bits, dy, dz_add, want_x, want_w, want_aff
        return None, gw1, None, gw2, None, gbar, None, None, None, zbar, None, dybar.view_as(t_dx), None, None, None, None



def _ffn_pair_enabled() -> bool:
    """options.ffn_pair = False: the node and the edge feed-forward of a block as two autograd nodes (equivalence tests)."""
    return options.ffn_pair


class _FFNLNPair(Function):
    """The two feed-forward halves of an Encoder_Block -- ``x = ln5(x + mlp(x))`` over the B N node rows and
    ``y = ln6(y + mlp2(y))`` over the B N^2 edge rows (reference layers.py:191-192) -- as ONE autograd node: each of
    its launches over the node rows rides in the launch of the same kernel over the edge rows (``_pair_launches``).
    Per branch exactly ``_FFNLN``: same kernels, same saved tensors, same extra outputs (pre-LayerNorm sum, row statistics)."""

    @staticmethod
    def forward(ctx, eps_n, eps_e, *args):      # args = (x, w1, b1, w2, b2, gamma, beta) of the node branch, then of the edge branch
        lib = _lib.load()
        keep = any(ctx.needs_input_grad)
        probs = []
        for inp, w1, b1, w2, b2, gamma, beta in (args[0:7], args[7:14]):
            H, C = w1.shape
            x2 = _c(inp).reshape(-1, C)
            R = x2.shape[0]
            dev, adt = x2.device, x2.dtype
            code = _hidden_code(adt)
            fused = fused_ffn_f32_supported(x2, w1, w2) and (not probs or probs[0]["fused"])
            if fused:      # (both problems or neither: one launch carries them)
                code = _lib.F32_H16
            elif probs and probs[0]["fused"]:
                probs[0]["fused"] = False
                probs[0]["code"] = _hidden_code(adt)
                probs[0]["h"] = _hidden_empty(probs[0]["R"], probs[0]["H"], adt, probs[0]["code"], dev)
            probs.append(dict(
                inp=inp, x2=x2, R=R, C=C, H=H, code=code, w1=w1, b1=b1, w2=w2, b2=b2, gamma=gamma, beta=beta, fused=fused,
                y=torch.empty(R, C, dtype=adt, device=dev),
                h=_hidden_empty(R, H, adt, code, dev) if (keep or not fused) else None,
                pre=torch.empty(R, C, dtype=adt, device=dev) if keep else None,
                mean=torch.empty(R, dtype=torch.float32, device=dev), rstd=torch.empty(R, dtype=torch.float32, device=dev),
                bits=torch.empty(int(lib.dg_row_gemm_mask_words(R, C, H, code)), dtype=torch.int32, device=dev) if keep else None))
        ref = probs[0]["x2"]
        fused = probs[0]["fused"] and probs[1]["fused"]
        cargs = []
        for p, eps in zip(probs, (eps_n, eps_e)):      # dg_ffn_fwd_args: h = relu(x W1^T + b1), y = LN(x + h W2^T + b2)
            p["eps"] = eps
            if fused:
                cargs.append(_ffn_f32_fwd_args(p, keep))
                continue
            cargs.append(_lib.FFNFwdArgs(
                _lib.ptr(p["x2"]), packed_weight(p["w1"], 0, ref.dtype).data_ptr(), _lib.fptr(_c(p["b1"])),
                packed_weight(p["w2"], 0, ref.dtype).data_ptr(), _lib.fptr(_c(p["b2"])), _lib.fptr(_c(p["gamma"])),
                _lib.fptr(_c(p["beta"])), _lib.ptr(p["y"]), _hptr(p["h"]), None if p["bits"] is None else p["bits"].data_ptr(),
                _lib.ptr(p["pre"]), _lib.ptr(p["mean"]), _lib.ptr(p["rstd"]), p["R"], float(eps)))
        with _dev(ref):
            if fused:      # ONE launch: the node rows ride in the launch over the edge rows, the hidden tensors stay on chip
                _lib.check(lib.dg_ffn_ln_fwd_f32(ctypes.byref(cargs[0]), ctypes.byref(cargs[1]), _lib.stream_of(ref)), "dg_ffn_ln_fwd_f32")
            else:          # one call: node, edge, node, edge inside dg_launch_pair_begin / _end
                _lib.check(lib.dg_edge_ffn_ln_fwd_pair(ctypes.byref(cargs[0]), ctypes.byref(cargs[1]), probs[0]["C"], probs[0]["H"],
                                                       probs[0]["code"], _lib.stream_of(ref)), "dg_edge_ffn_ln_fwd_pair")
        es = ref.element_size()
        for p in probs:
            R, C, H = p["R"], p["C"], p["H"]
            if fused:
                _account_ffn_f32(R, C, H, keep)
                continue
            hb = _hrow_bytes(p["code"], es, H)
            _account(_gemm_key(R, C, H), R * (es * C + hb), 2 * R * C * H)
            _account(_gemm_key(R, H, C), R * (hb + es * (3 if keep else 2) * C), 2 * R * C * H, floor=R * (hb + es * 2 * C))
        pn, pe = probs
        outs = (pn["y"].view(pn["inp"].shape), pn["pre"], pn["mean"], pn["rstd"],
                pe["y"].view(pe["inp"].shape), pe["pre"], pe["mean"], pe["rstd"])
        ctx.mark_non_differentiable(pn["mean"], pn["rstd"], pe["mean"], pe["rstd"])
        ctx.alias = False
        if not keep:
            return outs
        args = list(args)
        aliases = ()
        if in_second_order_forward() and _alias_outputs_enabled():
            # the penalty's forward: w1, w2, gamma of both branches leave as alias outputs (see _weight_alias)
            ctx.alias = True
            for i in (1, 3, 5, 8, 10, 12):
                args[i] = _weight_alias(args[i])
            aliases = tuple(args[i] for i in (1, 3, 5, 8, 10, 12))
        ctx.save_for_backward(*args[0:7], pn["h"], pn["mean"], pn["rstd"], pn["pre"], pn["bits"],
                              *args[7:14], pe["h"], pe["mean"], pe["rstd"], pe["pre"], pe["bits"])
        ctx.set_materialize_grads(False)
        return outs + aliases

    @staticmethod
    def backward(ctx, dyn, dpren, _dmn, _drn, dye, dpree, _dme=None, _dre=None, *galias):
        sv = ctx.saved_tensors
        call = []
        for base, off, dy, dpre in ((0, 2, dyn, dpren), (12, 9, dye, dpree)):
            x, w1, b1, w2, b2, gamma, beta, h, mean, rstd, pre, bits = sv[base:base + 12]
            want_w = ctx.needs_input_grad[off + 1] and not _inputs_only()
            want_aff = (ctx.needs_input_grad[off + 5] or ctx.needs_input_grad[off + 6]) and not _inputs_only()
            if dy is None and (dpre is None or torch.is_grad_enabled()):
                dy = torch.zeros_like(pre)
            call += [x, w1, b1, w2, b2, gamma, h, mean, rstd, pre, bits, dy, dpre, ctx.needs_input_grad[off], want_w, want_aff]
        o = list(_FFNLNPairBwd.apply(*call))
        if any(g is not None for g in galias):      # second-order gradients of w1, w2, gamma (node), w1, w2, gamma (edge)
            idx = (1, 3, 5, 8, 10, 12)
            for i, v in zip(idx, _join_alias_grads([o[i] for i in idx], galias)):
                o[i] = v
        return (None, None, *o[0:7], *o[7:14])


class _FFNLNPairBwd(Function):
    """Backward of ``_FFNLNPair`` as a differentiable node: per branch the sequence of ``_FFNLNBwd`` (LayerNorm backward,
    dh = (dz W2) * m, dx = dz + dh W1, the two weight gradients), node-level launches riding in the edge-level ones; its own
    backward (second order of the gradient penalty) pairs the same way."""

    @staticmethod
    def forward(ctx, *args):      # per branch: x, w1, b1, w2, b2, gamma, h, mean, rstd, pre, bits, dy, dz_add, want_x, want_w, want_aff
        lib = _lib.load()
        probs = []
        for x, w1, b1, w2, b2, gamma, h, mean, rstd, pre, bits, dy, dz_add, want_x, want_w, want_aff in (args[0:16], args[16:32]):
            H, C = w1.shape
            R = pre.shape[0]
            adt = pre.dtype
            p = dict(x=x, x2=_c(x).reshape(-1, C), w1=w1, w2=w2, gamma=gamma, h=h, mean=mean, rstd=rstd, pre=pre, bits=bits,
                     R=R, C=C, H=H, want_x=want_x, want_w=want_w, want_aff=want_aff, had_add=dz_add is not None,
                     dgamma=None, dbeta=None)
            if dy is None:      # dz_add IS the LayerNorm input gradient (made by dg_row_gemm_ln_bwd in the consumer of y)
                p["dy2"] = None
                p["dz"] = p["dz_add"] = _c(dz_add if dz_add.dtype == adt else dz_add.to(adt)).reshape(-1, C)
            else:
                p["dy2"] = _c(dy if dy.dtype == adt else dy.to(adt)).reshape(-1, C)
                p["dz_add"] = None if dz_add is None else _c(dz_add if dz_add.dtype == adt else dz_add.to(adt)).reshape(-1, C)
                p["dz"] = None
            probs.append(p)
        ref = probs[0]["pre"]
        adt, dev, es = ref.dtype, ref.device, ref.element_size()
        code, dh_code = _ffn_bwd_codes(probs[0]["h"], adt, probs[0]["R"], probs[0]["H"])
        cargs = []
        with _dev(ref):
            for i, p in enumerate(probs):      # dg_ffn_bwd_args: outputs and a workspace of its own per branch
                R, C, H = p["R"], p["C"], p["H"]
                if p["dy2"] is not None:
                    p["dz"] = torch.empty(R, C, dtype=adt, device=dev)
                    if p["want_aff"]:      # adjacent in memory: their reduction joins the call's single reduce launch
                        p["dgamma"], p["dbeta"] = torch.empty(2, p["gamma"].numel(), dtype=p["gamma"].dtype, device=dev).unbind(0)
                p["dh"] = _hidden_empty(R, H, adt, dh_code, dev)
                p["dx"] = torch.empty(R, C, dtype=adt, device=dev) if p["want_x"] else None
                p["dw1"] = p["db1"] = p["dw2"] = p["db2"] = None
                if p["want_w"]:
                    p["dw1"], p["dw2"] = torch.empty_like(p["w1"]), torch.empty_like(p["w2"])
                    p["db1"] = torch.empty(H, dtype=torch.float32, device=dev)
                    p["db2"] = torch.empty(C, dtype=torch.float32, device=dev)
                ws = _scratch(ref, int(lib.dg_edge_ffn_ln_workspace_bytes(R, C, H)), f"ffn_pair{i}")
                cargs.append(_lib.FFNBwdArgs(
                    _lib.ptr(p["x2"]), _hptr(p["h"]), p["bits"].data_ptr(), _lib.ptr(p["pre"]), _lib.ptr(p["mean"]),
                    _lib.ptr(p["rstd"]), _lib.fptr(_c(p["gamma"])), packed_weight(p["w1"], 1, adt).data_ptr(),
                    packed_weight(p["w2"], 1, adt).data_ptr(), _lib.ptr(p["dy2"]), _lib.ptr(p["dz_add"]), _lib.ptr(p["dz"]),
                    _hptr(p["dh"]), _lib.ptr(p["dx"]), _lib.ptr(p["dgamma"]), _lib.ptr(p["dbeta"]), _lib.ptr(p["dw1"]),
                    _lib.ptr(p["db1"]), _lib.ptr(p["dw2"]), _lib.ptr(p["db2"]), ws.data_ptr(), ws.numel(), R))
            # one call: LayerNorm backward, dh = (dz W2) * m, dx = dz + dh W1, dW2 = dz^T h, dW1 = dh^T x -- node, edge, node, edge
            # inside dg_launch_pair_begin / _end, one reduce launch for everything
            _lib.check(lib.dg_edge_ffn_ln_bwd_pair(ctypes.byref(cargs[0]), ctypes.byref(cargs[1]), probs[0]["C"], probs[0]["H"],
                                                   code, _lib.stream_of(ref)), "dg_edge_ffn_ln_bwd_pair")
        for p in probs:
            R, C, H = p["R"], p["C"], p["H"]
            if p["dy2"] is not None:
                _account("ln_bwd", es * R * C * (4 if p["dz_add"] is not None else 3))
            hb = _hrow_bytes(_hidden_code_of(p["h"], R, H) if _is_h16(p["h"]) else _lib.dt(ref), es, H)
            dhb = _hrow_bytes(dh_code, es, H)
            _account(_gemm_key(R, C, H), R * (es * C + dhb), 2 * R * C * H)
            if p["dx"] is not None:
                _account(_gemm_key(R, H, C), R * (dhb + es * 2 * C), 2 * R * C * H)
            if p["want_w"]:
                _account(_wgrad_key(R, C, H), R * (es * C + (2 * H + 4 if code == _lib.F32_H32_DH16 else hb)), 2 * R * C * H)
                _account(_wgrad_key(R, H, C), R * (es * C + dhb), 2 * R * C * H)
        saved, outs = [], []
        for p in probs:
            saved += [p["x"], p["w1"], p["w2"], p["gamma"], p["h"], p["mean"], p["rstd"], p["pre"], p["bits"], p["dy2"], p["dz"], p["dh"]]
            outs += [None if p["dx"] is None else p["dx"].view(p["x"].shape), p["dw1"], p["db1"], p["dw2"], p["db2"],
                     p["dgamma"], p["dbeta"]]
        ctx.save_for_backward(*saved)
        ctx.had_add = tuple(p["had_add"] for p in probs)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *t):
        for k in (0, 7):
            if any(g is not None for g in t[k + 1:k + 7]):
                raise RuntimeError("ffn_ln: second-order terms through parameter gradients are not implemented")
        sv = ctx.saved_tensors
        probs = []
        for i, k in enumerate((0, 7)):
            if t[k] is None:
                probs.append(None)
                continue
            if ctx.had_add[i]:
                raise RuntimeError("ffn_ln: third-order differentiation is not implemented")
            x, w1, w2, gamma, h, mean, rstd, pre, bits, dy2, dz, dh = sv[12 * i:12 * i + 12]
            H, C = w1.shape
            adt = pre.dtype
            probs.append(dict(w1=w1, w2=w2, gamma=gamma, mean=mean, rstd=rstd, pre=pre, bits=bits, dy2=dy2, dz=dz, dh=dh, C=C, H=H,
                              t_dx=t[k], t=_c(t[k] if t[k].dtype == adt else t[k].to(adt)).reshape(-1, C)))
        live = [p for p in probs if p is not None]
        if not live:
            return (None,) * 32
        ref = live[0]["pre"]
        pw = lambda w_, m_: packed_weight(w_, m_, ref.dtype)
        code = _hidden_code_of(live[0]["dh"], live[0]["t"].shape[0], live[0]["H"]) if _is_h16(live[0]["dh"]) else _lib.dt(ref)
        with _pair_launches(ref):
            for p in live:
                p["vbar"] = row_gemm(p["t"], pw(p["w1"], 0), p["C"], p["H"], mask_bits=p["bits"], code=code)   # (t W1^T) * m
            for p in live:
                p["ubar"] = row_gemm(p["vbar"], pw(p["w2"], 0), p["H"], p["C"], residual=p["t"], R=p["t"].shape[0])   # t + vbar W2^T
        for p in live:
            p["zbar"], p["dybar"], p["gbar"] = _ln_bwd2_rows(p["pre"], p["gamma"], p["mean"], p["rstd"], p["dy2"], p["ubar"])
            p["gw1"] = p["gw2"] = None
        if not _inputs_only():
            with _pair_launches(ref):
                res = _wgrad_many([(p["dh"], p["t"], False) for p in live] +          # ((u W2)*m)^T t
                                  [(p["dz"], p["vbar"], False) for p in live])        # u^T ((t W1^T)*m)
            for p, r in zip(live, res[:len(live)]):
                p["gw1"] = r[0]
            for p, r in zip(live, res[len(live):]):
                p["gw2"] = r[0]
        out = []
        for p in probs:
            if p is None:
                out += [None] * 16
            else:
                # inputs: x, w1, b1, w2, b2, gamma, h, mean, rstd, pre, bits, dy, dz_add, want_x, want_w, want_aff
                out += [None, p["gw1"], None, p["gw2"], None, p["gbar"], None, None, None, p["zbar"], None,
                        p["dybar"].view_as(p["t_dx"]), None, None, None, None]
        return tuple(out)


def ffn_ln_pair(x, node, y, edge):
    """``(ffn_ln(x, *node), ffn_ln(y, *edge, want_handle=True))`` -- node = (w1, b1, w2, b2, gamma, beta, eps) of mlp / ln5,
    edge the same of mlp2 / ln6 -- as one autograd node whose node-level launches ride in the edge-level ones
    (``_FFNLNPair``; float32 activations, dim 128, hidden 384).  Returns (x_out, y_out, LNHandle of ln6 or None)."""
    def ok(t, w1, b1, w2, b2):
        H, C = w1.shape
        return (t.is_cuda and t.dtype == torch.float32 and C == 128 and H == 384 and tuple(w2.shape) == (C, H)
                and b1 is not None and b2 is not None)
    if not (_ffn_pair_enabled() and ok(x, *node[:4]) and ok(y, *edge[:4]) and x.device == y.device):
        xo = ffn_ln(x, *node)
        yo, handle = ffn_ln(y, *edge, want_handle=True)
        return xo, yo, handle
    xo, _pn, _mn, _rn, yo, pre, mean, rstd = _FFNLNPair.apply(float(node[6]), float(edge[6]), x, *node[:6], y, *edge[:6])[:8]
    handle = LNHandle(pre, mean, rstd, edge[4], edge[5]) if (pre is not None and pre.requires_grad) else None
    return xo, yo, handle


_ffn_pack_cache = {}


def _ffn_packed_bf16(w1, w2):
    """The four bf16 fragment-order copies of (fc1.weight, fc2.weight) the fused bf16 feed-forward kernels keep
    in registers (dg_ffn_bf16_pack), cached like ``packed_weight``."""
    key = (id(w1), id(w2))
    hit = _ffn_pack_cache.get(key)
    if (hit is not None and hit[0]() is w1 and hit[1]() is w2 and hit[2] == (w1._version, w2._version)
            and hit[4] == (w1.data_ptr(), w2.data_ptr()) and hit[5] == _weights_epoch):
        return hit[3]
    if len(_ffn_pack_cache) > 1024:
        for k in [k for k, v in _ffn_pack_cache.items() if v[0]() is None or v[1]() is None]:
            del _ffn_pack_cache[k]
    lib = _lib.load()
    packed = torch.empty(int(lib.dg_ffn_bf16_packed_bytes()), dtype=torch.uint8, device=w1.device)
    with _dev(w1):
        _lib.check(lib.dg_ffn_bf16_pack(_lib.fptr(_c(w1.detach())), _lib.fptr(_c(w2.detach())), packed.data_ptr(),
                                        _lib.stream_of(w1)), "dg_ffn_bf16_pack")
    _ffn_pack_cache[key] = (weakref.ref(w1), weakref.ref(w2), (w1._version, w2._version), packed,
                            (w1.data_ptr(), w2.data_ptr()), _weights_epoch)
    return packed


class _FFNLNFusedBF16(Function):
    """LN(x + fc2(relu(fc1 x))) on the fused bf16 kernels (csrc/ffn_bf16.hip): the [R, 384] hidden tensor stays
    in LDS, the backward recomputes it (saved: pre-LayerNorm sum, mean / rstd, one ReLU bit per hidden element).
    First order only -- graphs that will be differentiated twice are built from ``_FFNLN`` (``ffn_ln`` below);
    if somebody differentiates this node twice anyway it falls back to the composite."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        R = x2.shape[0]
        lib = _lib.load()
        dev = x2.device
        record = any(ctx.needs_input_grad)
        Rp = int(lib.dg_ffn_bf16_padded_rows(R))      # the kernel stores whole 64-row tiles
        y = torch.empty(Rp, C, dtype=torch.bfloat16, device=dev)[:R]
        mean = torch.empty(Rp, dtype=torch.float32, device=dev)[:R]
        rstd = torch.empty(Rp, dtype=torch.float32, device=dev)[:R]
        pre = torch.empty(Rp, C, dtype=torch.bfloat16, device=dev)[:R] if record else None
        bits = torch.empty(int(lib.dg_ffn_bf16_mask_words(R)), dtype=torch.int32, device=dev) if record else None
        with _dev(x2):
            _lib.check(lib.dg_ffn_ln_fwd_bf16(_lib.ptr(x2), _ffn_packed_bf16(w1, w2).data_ptr(), _lib.fptr(_c(b1)),
                                              _lib.fptr(_c(b2)), _lib.fptr(_c(gamma)), _lib.fptr(_c(beta)), _lib.ptr(y),
                                              _lib.ptr(pre), _lib.ptr(mean), _lib.ptr(rstd),
                                              None if bits is None else bits.data_ptr(), R, eps, _lib.stream_of(x2)),
                       "dg_ffn_ln_fwd_bf16")
        _account("ffn" if R >= _lib.edge_rows() else "ffn_node", 2 * R * C * (3 if record else 2) + (48 * R if record else 0), 4 * R * C * 3 * C,
                 floor=2 * R * C * 2)
        if record:
            ctx.save_for_backward(x, w1, b1, w2, b2, gamma, beta, pre, mean, rstd, bits)
        ctx.eps = eps
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, b2, gamma, beta, pre, mean, rstd, bits = ctx.saved_tensors
        if torch.is_grad_enabled():
            eps = ctx.eps
            return _double_backward_fallback(lambda *t: _composite_ffn_ln(*t, eps), (x, w1, b1, w2, b2, gamma, beta),
                                             dy) + (None,)
        C, H = w1.shape[1], w1.shape[0]
        x2 = _c(x).reshape(-1, C)
        R = x2.shape[0]
        lib = _lib.load()
        dev = x2.device
        dy2 = _c(dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16)).reshape(-1, C)
        want_x = ctx.needs_input_grad[0]
        want_w = ctx.needs_input_grad[1] and not _inputs_only()
        dz = torch.empty(R, C, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(R, C, dtype=torch.bfloat16, device=dev) if want_x else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        dw1 = db1 = dw2 = db2 = bits2 = None
        if want_w:
            dw1, db1 = torch.empty_like(w1), torch.empty(H, dtype=torch.float32, device=dev)
            dw2, db2 = torch.empty_like(w2), torch.empty(C, dtype=torch.float32, device=dev)
            bits2 = torch.empty_like(bits)
        need = int(lib.dg_ffn_bf16_workspace_bytes(R))
        with _dev(x2):
            ws = _scratch(x2, need, "ffn16")
            _lib.check(lib.dg_ffn_ln_bwd_bf16(_lib.ptr(x2), _lib.ptr(pre), _lib.ptr(mean), _lib.ptr(rstd), bits.data_ptr(),
                                              _lib.fptr(_c(gamma)), _ffn_packed_bf16(w1, w2).data_ptr(), _lib.fptr(_c(b1)),
                                              _lib.ptr(dy2), _lib.ptr(dz), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                              _lib.ptr(dw1), _lib.ptr(db1), _lib.ptr(dw2), _lib.ptr(db2),
                                              None if bits2 is None else bits2.data_ptr(), ws.data_ptr(), ws.numel(), R,
                                              _lib.stream_of(x2)), "dg_ffn_ln_bwd_bf16")
        lvl = "" if R >= _lib.edge_rows() else "_node"
        _account("ffn" + lvl, 2 * R * C * (4 if want_x else 3) + 48 * R, 4 * R * C * H if want_x else 2 * R * C * H)
        if want_w:
            _account("ffn_wgrad" + lvl, 2 * (2 * R * C * 2 + 48 * R), 8 * R * C * H)
        if not ctx.needs_input_grad[5] or _inputs_only():
            dgamma = dbeta = None
        return (None if dx is None else dx.view(x.shape)), dw1, db1, dw2, db2, dgamma, dbeta, None


class LNHandle:
    """What the consumer of a LayerNorm output needs to run that LayerNorm's backward in the epilogue of its own
    input-gradient GEMM (dg_row_gemm_ln_bwd): the saved pre-LayerNorm sum (an autograd output of the producing node:
    the consumer returns dz as ITS gradient), the row statistics and the affine parameters."""
    __slots__ = ("pre", "mean", "rstd", "gamma", "beta")

    def __init__(self, pre, mean, rstd, gamma, beta):
        self.pre, self.mean, self.rstd, self.gamma, self.beta = pre, mean, rstd, gamma, beta


def ffn_ln(x, w1, b1, w2, b2, gamma, beta, eps: float = 1e-5, want_handle: bool = False):
    """LayerNorm(x + fc2(relu(fc1(x)))) with everything elementwise fused into the GEMM
    epilogues (dim 128, hidden 384); other shapes / second-order graphs use the composite.
    ``want_handle``: returns (y, LNHandle or None) -- see ``attn_block(y_ln=...)``."""
    H, C = w1.shape
    ok = (x.is_cuda and x.dtype in _lib.DTYPES and C == 128 and H == 384 and tuple(w2.shape) == (C, H)
          and b1 is not None and b2 is not None)
    handle = None
    if not ok:
        y = _composite_ffn_ln(x, w1, b1, w2, b2, gamma, beta, float(eps))
    elif x.dtype == torch.bfloat16 and not in_second_order_forward() and _fused_ffn_enabled():
        y = _FFNLNFusedBF16.apply(x, w1, b1, w2, b2, gamma, beta, float(eps))
    else:
        y, pre, mean, rstd = _FFNLN.apply(x, w1, b1, w2, b2, gamma, beta, float(eps))
        if want_handle and x.dtype == torch.float32 and pre is not None and pre.requires_grad:
            handle = LNHandle(pre, mean, rstd, gamma, beta)
    return (y, handle) if want_handle else y


def linear_relu(x, weight, bias):
    """relu(x W^T + b) (reference layers.py:50-51)."""
    return torch.relu(linear(x, weight, bias))


def _composite_linear_ln(x, w, b, residual, gamma, beta, eps):
    return ln_residual(residual, linear(x, w, b), gamma, beta, eps)


class _LinearLN(Function):
    @staticmethod
    def forward(ctx, x, w, b, residual, gamma, beta, eps):
        N, K = w.shape
        x2 = _c(x).reshape(-1, K)
        r2 = _c(residual).reshape(-1, N)
        y, mean, rstd, pre = row_gemm(x2, packed_weight(w, 0, x2.dtype), K, N, bias=b, residual=r2,
                                      ln=(_c(gamma), _c(beta), eps), want_pre=True)
        ctx.save_for_backward(x, w, b, residual, gamma, beta, mean, rstd, pre)
        ctx.eps = eps
        return y.view(residual.shape)

    @staticmethod
    def backward(ctx, dy):
        x, w, b, residual, gamma, beta, mean, rstd, pre = ctx.saved_tensors
        if torch.is_grad_enabled():
            eps = ctx.eps
            g = _double_backward_fallback(lambda *t: _composite_linear_ln(*t, eps),
                                          (x, w, b, residual, gamma, beta), dy)
            return g + (None,)
        N, K = w.shape
        dz, dgamma, dbeta = _ln_bwd_rows(pre, gamma, mean, rstd, _c(dy if dy.dtype == pre.dtype else dy.to(pre.dtype)).reshape(-1, N))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = row_gemm(dz, packed_weight(w, 1, dz.dtype), N, K).view(x.shape)
        if ctx.needs_input_grad[1] and not _inputs_only():
            dw, db = _wgrad(dz, _c(x).reshape(-1, K), b is not None)
        return dx, dw, db, dz.view(residual.shape), dgamma, dbeta, None


def linear_ln(x, weight, bias, residual, gamma, beta, eps: float = 1e-5):
    """LayerNorm(residual + x W^T + b) * gamma + beta in one kernel: out_e + ln4 and
    mlp2.fc2 + ln6 (reference layers.py:127,188,190,192) and their node twins."""
    if not _fusable(x, weight) or tuple(weight.shape) != (128, 128) or bias is None or in_second_order_forward():
        return _composite_linear_ln(x, weight, bias, residual, gamma, beta, float(eps))
    return _LinearLN.apply(x, weight, bias, residual, gamma, beta, float(eps))


# --------------------------------------------------------------------------
# whole attention half of an Encoder_Block as one autograd node
# --------------------------------------------------------------------------
def _composite_attn_block(x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, b3, g4, b4, alpha, eps3,
                          eps4, need_edge):
    q, k, v = linear(x1, wq, bq), linear(x1, wk, bk), linear(x1, wv, bv)
    e = linear(y, we, be)
    s, o = attn_core(q, k, v, e, alpha, need_s=need_edge)
    x2 = linear_ln(o, won, bon, x1, g3, b3, eps3)
    if not need_edge:
        return x2
    return x2, linear_ln(s, woe, boe, y, g4, b4, eps4)


def attn_half_f32_supported(yf, N: int, C: int) -> bool:
    """dg_attn_half_f32_fwd (e projection + attention core + out_e + residual + ln4 as one float32 launch) serves C = 128 and
    row groups of at most 96 neighbours (above 48: two stages per row group, online softmax across them);
    options.attn_half_f32 = "off" keeps the three launches, "n48" keeps them above 48 neighbours (A/B measurements)."""
    mode = options.attn_half_f32
    return yf.is_cuda and yf.dtype == torch.float32 and C == 128 and N <= (48 if mode == "n48" else 96) and mode != "off"


def attn_half_f32_bwd1_supported(dy2f, B: int, N: int, C: int, graph: bool = False) -> bool:
    """dg_attn_half_f32_bwd1 (ln4 backward + out_e input gradient + attention-core backward as one float32 launch) serves
    C = 128 and row groups of at most 48 neighbours; its workgroups walk whole molecules, so it needs a batch that fills the
    chip (B >= 128; options.attn_half_f32_bwd = "force" lifts that for tests, "off" keeps the two launches, "nograph" keeps
    them only for passes a second order differentiates)."""
    mode = options.attn_half_f32_bwd
    return (dy2f.is_cuda and dy2f.dtype == torch.float32 and C == 128 and N <= 48 and mode != "off"
            and (B >= 128 or mode == "force") and (not graph or mode != "nograph"))


class _AttnBlock(Function):
    """x2 = LN3(x1 + out_n(o)), y2 = LN4(y + out_e(s)) with (s, o) = attention(q(x1), k(x1), v(x1), e(y))
    -- reference layers.py:111-135 + 186-190 -- as ONE autograd node: every projection is a row-GEMM
    launch with its bias / residual / LayerNorm epilogue, and in the backward every gradient
    accumulation (y feeds e-proj and the ln4 residual; x1 feeds q, k, v and the ln3 residual) is the
    residual operand of the next GEMM's epilogue instead of a separate elementwise add."""

    @staticmethod
    def forward(ctx, x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, b3, g4, b4, alpha, eps3, eps4,
                need_edge, ppre=None, pmean=None, prstd=None, pgamma=None, pbeta=None):
        # ppre .. pbeta: LNHandle of the LayerNorm that produced y (or None): its backward can then run in the
        # epilogue of this node's dy GEMM, the result leaving as the gradient of `ppre` instead of `y`
        B, N, C = x1.shape
        x1f, yf = _c(x1).reshape(-1, C), _c(y).reshape(-1, C)
        adt = x1f.dtype
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        if lin3_supported(x1f, (wq, wk, wv)):      # q, k, v share their input: one launch
            q, k, v = lin3(x1f, (wq, wk, wv), (bq, bk, bv))
        else:
            q = row_gemm(x1f, pw(wq, 0), C, C, bias=bq)
            k = row_gemm(x1f, pw(wk, 0), C, C, bias=bk)
            v = row_gemm(x1f, pw(wv, 0), C, C, bias=bv)
        lib = _lib.load()
        # no input needs a gradient (the Generator's forward inside the D step): the pre-LayerNorm sums are not written
        keep = any(ctx.needs_input_grad)
        o = torch.empty_like(q)
        fused_edge = need_edge and attn_half_f32_supported(yf, N, C)
        if fused_edge:
            # e projection, scores, softmax / node output, out_e, residual, ln4: one launch (dg_attn_half_f32_fwd); e, s and
            # the pre-LayerNorm sum are written for the backward only
            R = yf.shape[0]
            dev = yf.device
            e = torch.empty(R, C, dtype=adt, device=dev) if keep else None
            s = torch.empty(R, C, dtype=adt, device=dev) if keep else None
            y2 = torch.empty(R, C, dtype=adt, device=dev)
            pre4 = torch.empty(R, C, dtype=adt, device=dev) if keep else None
            mean4 = torch.empty(R, dtype=torch.float32, device=dev)
            rstd4 = torch.empty(R, dtype=torch.float32, device=dev)
            with _dev(q):
                _lib.check(lib.dg_attn_half_f32_fwd(_lib.ptr(yf), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), pw(we, 0).data_ptr(),
                                                    _lib.fptr(_c(be)), pw(woe, 0).data_ptr(), _lib.fptr(_c(boe)), _lib.fptr(_c(g4)),
                                                    _lib.fptr(_c(b4)), _lib.ptr(e), _lib.ptr(s), _lib.ptr(o), _lib.ptr(y2),
                                                    _lib.ptr(pre4), _lib.ptr(mean4), _lib.ptr(rstd4), B, N, C, alpha, eps4,
                                                    _lib.stream_of(q)), "dg_attn_half_f32_fwd")
            _account("attn_half_fwd", 4 * (R * C * (5 if keep else 2) + 4 * B * N * C), 4 * R * C * C,
                     floor=4 * (R * C * 2 + 4 * B * N * C))
        else:
            e = row_gemm(yf, pw(we, 0), C, C, bias=be)
            s = torch.empty_like(e) if need_edge else None
            with _dev(q):
                _lib.check(lib.dg_attn_core_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(s),
                                                _lib.ptr(o), B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)), "dg_attn_core_fwd")
            _account("attn_fwd", q.element_size() * B * ((2 if need_edge else 1) * N * N * C + 4 * N * C))
        r3 = row_gemm(o, pw(won, 0), C, C, bias=bon, residual=x1f, ln=(_c(g3), _c(b3), eps3), want_pre=keep)
        x2, mean3, rstd3, pre3 = r3 if keep else (*r3, None)
        outs = [x2.view(B, N, C)]
        # the penalty's forward: the parameters leave as alias outputs (see _weight_alias)
        ctx.alias = bool(keep and in_second_order_forward() and _alias_outputs_enabled())
        aliases = ()
        if ctx.alias:
            aliases = tuple(_weight_alias(t) for t in (wq, wk, wv, we, woe, won, g3, g4))
            wq, wk, wv, we, woe, won, g3, g4 = aliases
        saved = [x1, y, wq, wk, wv, we, woe, won, g3, g4, q, k, v, e, s, o, mean3, rstd3, pre3]
        if need_edge:
            if not fused_edge:
                r4 = row_gemm(s, pw(woe, 0), C, C, bias=boe, residual=yf, ln=(_c(g4), _c(b4), eps4), want_pre=keep)
                y2, mean4, rstd4, pre4 = r4 if keep else (*r4, None)
            outs.append(y2.view(B, N, N, C))
            saved += [mean4, rstd4, pre4]
        else:
            pre4 = None
        ctx.has_prev = ppre is not None
        if ctx.has_prev:
            saved += [ppre, pmean, prstd, pgamma]
        ctx.save_for_backward(*saved)
        ctx.cfg = (alpha, eps3, eps4, need_edge, (B, N, C))
        ctx.extra = (bq, bk, bv, be, boe, bon, b3, b4)
        ctx.set_materialize_grads(False)
        # pre3 / pre4 / q / k / v / e are outputs only so that the second order of the gradient penalty
        # can return their adjoints to THIS node, where they join the first-order gradients inside one
        # backward pass (see _AttnBlockBwd.backward); module code never sees them.
        return tuple(outs) + ((pre3, pre4) if need_edge else (pre3,)) + (q, k, v, e) + aliases

    @staticmethod
    def backward(ctx, dx2, *more):
        alpha, eps3, eps4, need_edge, (B, N, C) = ctx.cfg
        galias = ()
        if ctx.alias:      # second-order gradients of wq, wk, wv, we, woe, won, g3, g4 (or None each)
            more, galias = more[:-8], more[-8:]
        sv = ctx.saved_tensors
        x1, y, wq, wk, wv, we, woe, won, g3, g4, q, k, v, e, s, o, mean3, rstd3, pre3 = sv[:19]
        mean4, rstd4, pre4 = sv[19:22] if need_edge else (None, None, None)
        bq, bk, bv, be, boe, bon, b3, b4 = ctx.extra
        if need_edge:
            dy2, add3, add4, aq, ak, av, ae = more
            if dy2 is None:
                dy2 = torch.zeros_like(pre4)
        else:
            dy2 = add4 = None
            add3, aq, ak, av, ae = more
        if dx2 is None:
            dx2 = torch.zeros_like(pre3)
        wants_w = ctx.needs_input_grad[2] and not _inputs_only()
        want_aff = any(ctx.needs_input_grad[14:18]) and not _inputs_only()      # ln3 / ln4 affine parameters
        ppre = pmean = prstd = pgamma = None
        if ctx.has_prev:
            ppre, pmean, prstd, pgamma = sv[-4:]
        # y is the output of a LayerNorm whose handle came with it, and no graph is being recorded: that LayerNorm's
        # backward runs as the epilogue of the dy GEMM (its result is the gradient of `ppre`, y itself gets none)
        # (not in the last pass of a double backward -- recognisable by the adjoints of this node's extra outputs: there
        # the producing feed-forward node's `pre` ALSO receives the second-order adjoint, and autograd would join the
        # two with an edge-level add that costs more than the fused LayerNorm backward saves)
        second_pass = any(t is not None for t in (add3, add4, aq, ak, av, ae))
        fuse_prev = bool(ctx.has_prev and not torch.is_grad_enabled() and not second_pass and ctx.needs_input_grad[1]
                         and ctx.needs_input_grad[22] and row_gemm_ln_bwd_supported(q, C)
                         and tuple(ppre.shape) == (B * N * N, C))
        outs = _AttnBlockBwd.apply(x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, g4,
                                   q, k, v, e, s, o, mean3, rstd3, pre3, mean4, rstd4, pre4, dx2, dy2,
                                   add3, add4, aq, ak, av, ae,
                                   alpha, need_edge, ctx.needs_input_grad[0], ctx.needs_input_grad[1], wants_w,
                                   ppre if fuse_prev else None, pmean, prstd, pgamma, want_aff,
                                   not torch.is_grad_enabled())      # (no graph is being recorded: see _AttnBlockBwd)
        (dx1, dy, dwq, dbq, dwk, dbk, dwv, dbv, dwe, dbe, dwoe, dboe, dwon, dbon, dg3, db3, dg4, db4, dzp, dgp, dbp) = outs
        if any(g is not None for g in galias):
            dwq, dwk, dwv, dwe, dwoe, dwon, dg3, dg4 = _join_alias_grads((dwq, dwk, dwv, dwe, dwoe, dwon, dg3, dg4), galias)
        if not (ctx.needs_input_grad[25] and not _inputs_only()):
            dgp = dbp = None
        return (dx1, dy, dwq, dbq, dwk, dbk, dwv, dbv, dwe, dbe, dwoe, dboe, dwon, dbon, dg3, db3, dg4, db4,
                None, None, None, None, dzp, None, None, dgp, dbp)


def _attn_bwd_launch(q, k, v, e, ws, wo, alpha, add_e=None):
    B, N, C = q.shape[0], q.shape[1], q.shape[2]
    lib = _lib.load()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    de = torch.empty_like(e)
    with _dev(q):
        _lib.check(lib.dg_attn_core_bwd_add(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(ws),
                                            _lib.ptr(wo), _lib.ptr(add_e), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv),
                                            _lib.ptr(de), B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)),
                   "dg_attn_core_bwd")
    _account("attn_bwd", q.element_size() * B * ((2 + (ws is not None) + (add_e is not None)) * N * N * C + 7 * N * C))
    return dq, dk, dv, de


def _attn_bwd2_launch(q, k, v, e, ws, wo, tq, tk, tv, te, alpha):
    B, N, C = q.shape[0], q.shape[1], q.shape[2]
    lib = _lib.load()
    gq, gk, gv, gwo = (torch.empty_like(q) for _ in range(4))
    ge = torch.empty_like(e)
    gws = torch.empty_like(e) if ws is not None else None
    with _dev(q):
        _lib.check(lib.dg_attn_core_bwd2(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(ws), _lib.ptr(wo),
                                         _lib.ptr(tq), _lib.ptr(tk), _lib.ptr(tv), _lib.ptr(te), _lib.ptr(gq),
                                         _lib.ptr(gk), _lib.ptr(gv), _lib.ptr(ge), _lib.ptr(gws), _lib.ptr(gwo),
                                         B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)), "dg_attn_core_bwd2")
    _account("attn_bwd2", q.element_size() * B * ((5 if ws is not None else 3) * N * N * C + 11 * N * C))
    return gq, gk, gv, ge, gws, gwo


class _AttnBlockBwd(Function):
    """Backward of ``_AttnBlock`` as a differentiable node; its own backward (second order of the
    gradient penalty) chains the same kernels: row GEMMs for every projection (forward packs where the
    first backward used the input-gradient packs and vice versa), dg_attn_core_bwd2 for the attention
    core, dg_ln_residual_bwd2 for ln3 / ln4.  The adjoints that reach the forward intermediates
    (pre-LayerNorm sums, q, k, v, e) are handed to the forward node as gradients of its extra outputs;
    they come back in as add3 / add4 / aq / ak / av / ae and are summed into the single first-order
    pass of that node."""

    @staticmethod
    def forward(ctx, *args):
        # one reduce launch for the block: six weight gradients + two LayerNorms' dgamma / dbeta
        wants_w = args[40] or (len(args) > 45 and args[45])      # weight gradients or LayerNorm affine gradients
        with _reduce_batch(args[0], on=bool(wants_w)) as inb:
            return _AttnBlockBwd._forward(ctx, inb, *args)

    @staticmethod
    def _forward(ctx, inb, x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, g4, q, k, v, e, s, o,
                 mean3, rstd3, pre3, mean4, rstd4, pre4, dx2, dy2, add3, add4, aq, ak, av, ae,
                 alpha, need_edge, want_x, want_y, wants_w, ppre=None, pmean=None, prstd=None, pgamma=None, want_aff=None,
                 no_graph=False):
        if want_aff is None:
            want_aff = wants_w
        B, N, C = x1.shape
        adt = q.dtype
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        cast = lambda t: t if t.dtype == adt else t.to(adt)
        x1f, yf = _c(x1).reshape(-1, C), _c(y).reshape(-1, C)
        dx2f = _c(cast(dx2)).reshape(-1, C)
        cadd = lambda t: None if t is None else _c(cast(t)).reshape(-1, C)
        dz3, dg3, db3 = _ln_bwd_rows(pre3, g3, mean3, rstd3, dx2f, cadd(add3), want_affine=want_aff,
                                     batch_slot=0 if inb else None)
        do = row_gemm(dz3, pw(won, 1), C, C).view(B, N, C)
        ds = dz4 = dg4 = db4 = dy2f = None
        qv, kv, vv, ev = q.view(B, N, C), k.view(B, N, C), v.view(B, N, C), e.view(B, N, N, C)
        fused1 = None
        if need_edge:
            dy2f = _c(cast(dy2)).reshape(-1, C)
            if (add4 is None and all(t is None for t in (aq, ak, av, ae))
                    and attn_half_f32_bwd1_supported(dy2f, B, N, C, graph=not no_graph)):
                # ln4 backward + ds = dz4 Woe + the attention core's backward: one launch; ds stays on chip unless a graph is
                # being recorded (the penalty's first backward: its second order reads ds)
                lib = _lib.load()
                dev = dy2f.device
                dz4, de = torch.empty_like(dy2f), torch.empty_like(dy2f)
                ds = None if no_graph else torch.empty_like(dy2f).view(B, N, N, C)
                dq, dk, dv = (torch.empty(B, N, C, dtype=adt, device=dev) for _ in range(3))
                dg4, db4 = (torch.empty(2, C, dtype=torch.float32, device=dev).unbind(0) if want_aff else (None, None))
                with _dev(dy2f):
                    ws = _scratch(dy2f, int(lib.dg_attn_half_f32_bwd1_workspace_bytes(B)), "ahb_batch" if inb else "ahb")
                    _lib.check(lib.dg_attn_half_f32_bwd1(_lib.ptr(dy2f), _lib.ptr(pre4), _lib.ptr(mean4), _lib.ptr(rstd4),
                                                         _lib.fptr(_c(g4)), pw(woe, 1).data_ptr(), _lib.ptr(ev), _lib.ptr(qv),
                                                         _lib.ptr(kv), _lib.ptr(vv), _lib.ptr(do), _lib.ptr(dz4), _lib.ptr(ds), _lib.ptr(de),
                                                         _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(dg4), _lib.ptr(db4),
                                                         ws.data_ptr(), ws.numel(), B, N, C, alpha, _lib.stream_of(dy2f)),
                               "dg_attn_half_f32_bwd1")
                _account("attn_half_bwd", 4 * (dy2f.shape[0] * C * (5 if ds is None else 6) + 7 * B * N * C), 2 * dy2f.shape[0] * C * C)
                de = de.view(B, N, N, C)
                fused1 = (dq, dk, dv, de)
            elif add4 is None and dy2f.shape[0] >= _lib.edge_rows() and ln_bwd_row_gemm_supported(dy2f, C, C):
                # ln4's backward runs in the producer waves of the out_e input-gradient GEMM (edge-level launches only:
                # at node level the three small launches it replaces are faster)
                dz4, ds, dg4, db4 = ln_bwd_row_gemm(pre4, g4, mean4, rstd4, dy2f, pw(woe, 1), want_affine=want_aff,
                                                    batch_slot=1 if inb else None)
                ds = ds.view(B, N, N, C)
            else:
                dz4, dg4, db4 = _ln_bwd_rows(pre4, g4, mean4, rstd4, dy2f, cadd(add4), want_affine=want_aff,
                                             batch_slot=1 if inb else None)
                ds = row_gemm(dz4, pw(woe, 1), C, C).view(B, N, N, C)
        # fp32: the adjoint of e joins de inside the kernel (one read stream instead of a 3-pass add).  The bf16
        # variant of that kernel is latency-bound at 2 waves / SIMD and the extra operand set costs more than the add
        # it saves (configs[2], A/B on one box: 217.2 vs 213.7 ms per step): bf16 adds afterwards.
        fold = ae is not None and adt == torch.float32
        aef = _c(cast(ae)).view(B, N, N, C) if fold else None      # joins de inside the kernel
        dq, dk, dv, de = fused1 if fused1 is not None else _attn_bwd_launch(qv, kv, vv, ev, ds, do, alpha, add_e=aef)
        pairs = [(got, extra.view(got.shape)) for got, extra in ((dq, aq), (dk, ak), (dv, av)) if extra is not None]
        if pairs:      # the node-level adjoints of q, k, v (second pass of the penalty): one multi-tensor launch
            torch._foreach_add_([g_ for g_, _ in pairs], [e_ if e_.dtype == g_.dtype else e_.to(g_.dtype) for g_, e_ in pairs])
        if ae is not None and not fold:
            de.add_(ae.view(de.shape))
        ctx.third = any(t is not None for t in (add3, add4, aq, ak, av, ae))
        dqf, dkf, dvf, def_ = dq.view(-1, C), dk.view(-1, C), dv.view(-1, C), de.view(-1, C)
        dy = dx1 = dzp = dgp = dbp = None
        if want_y and ppre is not None:      # + ln4 residual path, then the backward of the LayerNorm that made y
            dzp, dgp, dbp = row_gemm_ln_bwd(def_, pw(we, 1), C, dz4, ppre, pgamma, pmean, prstd)
        elif want_y:
            dy = row_gemm(def_, pw(we, 1), C, C, residual=dz4).view(y.shape)      # + ln4 residual path
        use3 = lin3_supported(dqf, (wq, wk, wv))      # dq Wq + dk Wk + dv Wv and the three weight gradients: one launch each
        if want_x and use3:
            dx1 = sum3(dqf, dkf, dvf, (wq, wk, wv), residual=dz3).view(x1.shape)   # + ln3 residual path
        elif want_x:
            t = row_gemm(dqf, pw(wq, 1), C, C, residual=dz3)                       # + ln3 residual path
            t = row_gemm(dkf, pw(wk, 1), C, C, residual=t)
            dx1 = row_gemm(dvf, pw(wv, 1), C, C, residual=t).view(x1.shape)
        gw = [None] * 12
        if wants_w:
            qkv_items = [((dqf, dkf, dvf), x1f, True)] if use3 else [(dqf, x1f, True), (dkf, x1f, True), (dvf, x1f, True)]
            items = qkv_items + [(def_, yf, True), (dz3, o, True)]
            if need_edge:
                items.append((dz4, s, True))
            # (out_n's weight gradient over the node rows rides in out_e's over the edge rows)
            res = _wgrad_many(items, open_batch=not inb, pair_from=len(items) - 2 if need_edge else None)
            if use3:      # rows 0..127 / 128..255 / 256..383 of the stacked gradient
                (w3, b3), res = res[0], res[1:]
                gw[0:6] = [w3[0:128], b3[0:128], w3[128:256], b3[128:256], w3[256:384], b3[256:384]]
            else:
                (gw[0], gw[1]), (gw[2], gw[3]), (gw[4], gw[5]) = res[:3]
                res = res[3:]
            (gw[6], gw[7]), (gw[10], gw[11]) = res[:2]
            if need_edge:
                gw[8], gw[9] = res[2]
        ctx.save_for_backward(x1, y, wq, wk, wv, we, woe, won, g3, g4, q, k, v, e, s, o, mean3, rstd3, pre3,
                              mean4, rstd4, pre4, dx2f, dy2f, dz3, dz4, do, ds, dq, dk, dv, de)
        ctx.cfg = (alpha, need_edge, (B, N, C), dx2.shape, None if dy2 is None else dy2.shape)
        ctx.set_materialize_grads(False)
        return (dx1, dy, *gw, dg3, db3, dg4, db4, dzp, dgp, dbp)

    @staticmethod
    @once_differentiable
    def backward(ctx, t1, ty, *rest):
        if any(r is not None for r in rest):
            raise RuntimeError("attn_block: second-order terms through parameter gradients are not implemented")
        if ctx.third:
            raise RuntimeError("attn_block: third-order differentiation is not implemented")
        alpha, need_edge, (B, N, C), dx2_shape, dy2_shape = ctx.cfg
        (x1, y, wq, wk, wv, we, woe, won, g3, g4, q, k, v, e, s, o, mean3, rstd3, pre3, mean4, rstd4, pre4,
         dx2f, dy2f, dz3, dz4, do, ds, dq, dk, dv, de) = ctx.saved_tensors
        adt = q.dtype
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        cast = lambda t: t if t.dtype == adt else t.to(adt)
        with_w = not _inputs_only()
        x1f, yf = _c(x1).reshape(-1, C), _c(y).reshape(-1, C)
        RN, RE = B * N, B * N * N
        zn = lambda: torch.zeros(RN, C, dtype=adt, device=q.device)
        t1f = _c(cast(t1)).reshape(-1, C) if t1 is not None else zn()
        tyf = _c(cast(ty)).reshape(-1, C) if ty is not None else torch.zeros(RE, C, dtype=adt, device=q.device)
        dqf, dkf, dvf, def_ = dq.view(-1, C), dk.view(-1, C), dv.view(-1, C), de.view(-1, C)
        # adjoints of dq, dk, dv, de (dx1 = dz3 + dq Wq + dk Wk + dv Wv ; dy = dz4 + de We)
        use3 = lin3_supported(t1f, (wq, wk, wv))
        if use3:
            tq, tk, tv = lin3(t1f, (wq, wk, wv), (None, None, None))
        else:
            tq = row_gemm(t1f, pw(wq, 0), C, C)
            tk = row_gemm(t1f, pw(wk, 0), C, C)
            tv = row_gemm(t1f, pw(wv, 0), C, C)
        te = row_gemm(tyf, pw(we, 0), C, C)
        qv, kv, vv, ev = q.view(B, N, C), k.view(B, N, C), v.view(B, N, C), e.view(B, N, N, C)
        gq, gk, gv, ge, gws, gwo = _attn_bwd2_launch(qv, kv, vv, ev, ds, do, tq.view(B, N, C), tk.view(B, N, C),
                                                     tv.view(B, N, C), te.view(B, N, N, C), alpha)
        # adjoints of dz3 / dz4 (do = dz3 Won, ds = dz4 Woe, plus the direct residual terms)
        adz3 = row_gemm(gwo.view(-1, C), pw(won, 0), C, C, residual=t1f)
        z3bar, dx2bar, g3bar = _ln_bwd2_rows(pre3, g3, mean3, rstd3, dx2f, adz3)
        z4bar = dy2bar = g4bar = None
        if need_edge:
            adz4 = row_gemm(gws.view(-1, C), pw(woe, 0), C, C, residual=tyf)
            z4bar, dy2bar, g4bar = _ln_bwd2_rows(pre4, g4, mean4, rstd4, dy2f, adz4)
        gW = [None] * 12
        if with_w:
            qkv_items = [((dqf, dkf, dvf), t1f, False)] if use3 else [(dqf, t1f, False), (dkf, t1f, False), (dvf, t1f, False)]
            items = qkv_items + [(def_, tyf, False), (dz3, gwo.view(-1, C), False)]
            if need_edge:
                items.append((dz4, gws.view(-1, C), False))
            res = _wgrad_many(items, pair_from=len(items) - 2 if need_edge else None)
            if use3:
                w3, res = res[0][0], res[1:]
                gW[0], gW[2], gW[4] = w3[0:128], w3[128:256], w3[256:384]
            else:
                gW[0], gW[2], gW[4] = (r[0] for r in res[:3])
                res = res[3:]
            gW[6], gW[10] = res[0][0], res[1][0]
            if need_edge:
                gW[8] = res[2][0]
        # The outputs depend on x1 / y only through the forward intermediates: their adjoints
        # (z3bar, z4bar at the pre-LayerNorm sums; gq, gk, gv, ge) go to the forward node.
        # inputs: x1, y, wq,bq, wk,bk, wv,bv, we,be, woe,boe, won,bon, g3, g4, q,k,v,e, s,o,
        #         mean3,rstd3,pre3, mean4,rstd4,pre4, dx2, dy2, 6 adds, 5 flags, 4 LNHandle fields, want_aff
        return (None, None, *gW, g3bar, g4bar, gq.view_as(q), gk.view_as(k), gv.view_as(v), ge.view_as(e), None, None,
                None, None, z3bar, None, None, z4bar, dx2bar.view(dx2_shape),
                None if dy2bar is None else dy2bar.view(dy2_shape), *([None] * 17))


_half_pack_cache = {}


def _attn_half_packed(we, woe, dtype):
    """Fragment-order copies of (e.weight, out_e.weight) and their transposes for the fused attention-half kernels
    (dg_attn_half_pack), cached like ``packed_weight``."""
    key = (id(we), id(woe), dtype)
    hit = _half_pack_cache.get(key)
    if (hit is not None and hit[0]() is we and hit[1]() is woe and hit[2] == (we._version, woe._version)
            and hit[4] == (we.data_ptr(), woe.data_ptr()) and hit[5] == _weights_epoch):
        return hit[3]
    if len(_half_pack_cache) > 1024:
        for k in [k for k, v in _half_pack_cache.items() if v[0]() is None or v[1]() is None]:
            del _half_pack_cache[k]
    lib = _lib.load()
    code = _lib.DTYPES[dtype]
    packed = torch.empty(int(lib.dg_attn_half_packed_bytes(code)), dtype=torch.uint8, device=we.device)
    with _dev(we):
        _lib.check(lib.dg_attn_half_pack(_lib.fptr(_c(we.detach())), _lib.fptr(_c(woe.detach())), packed.data_ptr(), code,
                                         _lib.stream_of(we)), "dg_attn_half_pack")
    _half_pack_cache[key] = (weakref.ref(we), weakref.ref(woe), (we._version, woe._version), packed,
                             (we.data_ptr(), woe.data_ptr()), _weights_epoch)
    return packed


def _fused_attn_half_enabled() -> bool:
    """options.attn_half = "unfused" keeps the bf16 attention half on the separate launches (A/B measurements)."""
    return options.attn_half != "unfused"


def attn_half_supported(dtype, N: int, C: int) -> bool:
    """Shapes the module path routes to the fused attention-half kernels.  The kernels accept N <= 96, but above 48 the
    backward keeps 6 row blocks of accumulators per lane and spills (N = 90, B = 64: 633 vs 645 molecules/s for the
    separate launches), so BASELINE configs[4] stays on those; options.attn_half = "force" routes every N <= 96 (tests)."""
    limit = 96 if options.attn_half == "force" else 48
    return dtype == torch.bfloat16 and C == 128 and 1 <= N <= limit


class _AttnBlockFused(Function):
    """The same block as ``_AttnBlock`` with the whole EDGE side -- e-projection, Hadamard score, softmax over j, AV,
    out_e, residual, ln4 (reference layers.py:116-135,186-190) -- in ONE kernel per direction (csrc/attn_half.hip):
    ``e`` and ``s`` never exist in HBM, the backward recomputes them from the saved layer input ``y`` and accumulates
    the weight gradients of e / out_e inside the kernel.  The node side (q, k, v, out_n + ln3; R = B N rows) stays on
    the row GEMMs.  First order only: graphs that will be differentiated twice are built from ``_AttnBlock``."""

    @staticmethod
    def forward(ctx, x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, b3, g4, b4, alpha, eps3, eps4,
                need_edge):
        B, N, C = x1.shape
        x1f = _c(x1).reshape(-1, C)
        yc = _c(y)
        adt = x1f.dtype
        code = _lib.dt(x1f)
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        q = row_gemm(x1f, pw(wq, 0), C, C, bias=bq)
        k = row_gemm(x1f, pw(wk, 0), C, C, bias=bk)
        v = row_gemm(x1f, pw(wv, 0), C, C, bias=bv)
        lib = _lib.load()
        dev = x1f.device
        o = torch.empty_like(q)
        y2 = pre4 = mean4 = rstd4 = None
        if need_edge:
            y2 = torch.empty_like(yc)
            pre4 = torch.empty_like(yc)
            mean4 = torch.empty(B * N * N, dtype=torch.float32, device=dev)
            rstd4 = torch.empty(B * N * N, dtype=torch.float32, device=dev)
        packed = _attn_half_packed(we, woe, adt)
        with _dev(q):
            _lib.check(lib.dg_attn_half_fwd(_lib.ptr(yc), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), packed.data_ptr(),
                                            _lib.fptr(_c(be)), _lib.fptr(_c(boe)), _lib.fptr(_c(g4)), _lib.fptr(_c(b4)),
                                            _lib.ptr(o), _lib.ptr(y2), _lib.ptr(pre4), _lib.ptr(mean4), _lib.ptr(rstd4),
                                            B, N, C, alpha, eps4, code, _lib.stream_of(q)), "dg_attn_half_fwd")
        es = q.element_size()
        _account("attn_half_fwd", es * B * ((3 if need_edge else 1) * N * N * C + 4 * N * C),
                 2 * B * N * N * C * C * (2 if need_edge else 1), floor=es * B * ((2 if need_edge else 1) * N * N * C + 4 * N * C))
        x2, mean3, rstd3, pre3 = row_gemm(o, pw(won, 0), C, C, bias=bon, residual=x1f, ln=(_c(g3), _c(b3), eps3),
                                          want_pre=True)
        ctx.save_for_backward(x1, yc, wq, wk, wv, we, woe, won, g3, g4, be, q, k, v, o, mean3, rstd3, pre3, mean4, rstd4,
                              pre4)
        ctx.cfg = (alpha, need_edge, (B, N, C))
        ctx.extra = (bq, bk, bv, boe, bon, b3, b4, eps3, eps4)
        ctx.set_materialize_grads(False)
        if need_edge:
            return x2.view(B, N, C), y2
        return x2.view(B, N, C)

    @staticmethod
    def backward(ctx, dx2, dy2=None):
        wants_w = bool((ctx.needs_input_grad[2] or any(ctx.needs_input_grad[14:18])) and not _inputs_only()
                       and not torch.is_grad_enabled())
        with _reduce_batch(ctx.saved_tensors[0], on=wants_w) as inb:      # one reduce launch for the block
            return _AttnBlockFused._backward(ctx, inb, dx2, dy2)

    @staticmethod
    def _backward(ctx, inb, dx2, dy2=None):
        alpha, need_edge, (B, N, C) = ctx.cfg
        (x1, y, wq, wk, wv, we, woe, won, g3, g4, be, q, k, v, o, mean3, rstd3, pre3, mean4, rstd4,
         pre4) = ctx.saved_tensors
        if torch.is_grad_enabled():      # create_graph=True outside second_order_forward(): composite graph
            bq, bk, bv, boe, bon, b3, b4, eps3, eps4 = ctx.extra
            ins = (x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, b3, g4, b4)
            gouts = (dx2, dy2) if need_edge else dx2
            return _double_backward_fallback(
                lambda *t: _composite_attn_block(*t, alpha, eps3, eps4, need_edge), ins, gouts) + (None,) * 4
        adt = q.dtype
        code = _lib.dt(q)
        dev = q.device
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        cast = lambda t: t if t.dtype == adt else t.to(adt)
        wants_w = ctx.needs_input_grad[2] and not _inputs_only()
        want_aff = any(ctx.needs_input_grad[14:18]) and not _inputs_only()
        x1f = _c(x1).reshape(-1, C)
        if dx2 is None:
            dx2 = torch.zeros_like(pre3)
        dz3, dg3, db3 = _ln_bwd_rows(pre3, g3, mean3, rstd3, _c(cast(dx2)).reshape(-1, C), want_affine=want_aff,
                                     batch_slot=0 if inb else None)
        do = row_gemm(dz3, pw(won, 1), C, C)
        dz4 = dg4 = db4 = None
        if need_edge:
            if dy2 is None:
                dy2 = torch.zeros_like(pre4)
            dz4, dg4, db4 = _ln_bwd_rows(pre4.view(-1, C), g4, mean4, rstd4, _c(cast(dy2)).reshape(-1, C), want_affine=want_aff,
                                         batch_slot=1 if inb else None)
        lib = _lib.load()
        dy = torch.empty_like(y)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        dwe = dbe = dwoe = dboe = None
        if wants_w:
            dwe = torch.empty_like(we)
            dbe = torch.empty(C, dtype=torch.float32, device=dev)
            if need_edge:
                dwoe = torch.empty_like(woe)
                dboe = torch.empty(C, dtype=torch.float32, device=dev)
        need = int(lib.dg_attn_half_bwd_workspace_bytes(B, N))
        with _dev(q):
            ws = _scratch(q, need, "half")
            _lib.check(lib.dg_attn_half_bwd(_lib.ptr(y), _lib.ptr(dz4), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(do),
                                            _attn_half_packed(we, woe, adt).data_ptr(), _lib.fptr(_c(be)), _lib.ptr(dy),
                                            _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(dwe), _lib.ptr(dbe),
                                            _lib.ptr(dwoe), _lib.ptr(dboe), ws.data_ptr(), ws.numel(), B, N, C, alpha,
                                            code, _lib.stream_of(q)), "dg_attn_half_bwd")
        es = q.element_size()
        _account("attn_half_bwd", es * B * ((3 if need_edge else 2) * N * N * C + 8 * N * C),
                 2 * B * N * N * C * C * ((3 if need_edge else 2) + (2 if wants_w and need_edge else (1 if wants_w else 0))))
        dx1 = None
        if ctx.needs_input_grad[0]:
            t = row_gemm(dq, pw(wq, 1), C, C, residual=dz3)                        # + ln3 residual path
            t = row_gemm(dk, pw(wk, 1), C, C, residual=t)
            dx1 = row_gemm(dv, pw(wv, 1), C, C, residual=t).view(x1.shape)
        gw = [None] * 12
        if wants_w:
            (gw[0], gw[1]), (gw[2], gw[3]), (gw[4], gw[5]), (gw[10], gw[11]) = _wgrad_many(
                [(dq, x1f, True), (dk, x1f, True), (dv, x1f, True), (dz3, o, True)], open_batch=not inb)
            gw[6], gw[7] = dwe, dbe
            gw[8], gw[9] = dwoe, dboe
        return (dx1, (dy if ctx.needs_input_grad[1] else None), *gw, dg3, db3, dg4, db4, None, None, None, None)


def attn_block(x1, y, attn, ln3, ln4, need_edge=True, y_ln=None):
    """Attention half of an encoder block for ``attn`` (an MHA module): returns
    (LN3(x1 + out_n(o)), LN4(y + out_e(s)) or None).  ``y_ln``: the LNHandle of the LayerNorm whose output y is
    (``ffn_ln(..., want_handle=True)``), or None."""
    C = x1.shape[-1]
    alpha = 1.0 / (attn.d_k ** 0.5)
    args = (x1, y, attn.q.weight, attn.q.bias, attn.k.weight, attn.k.bias, attn.v.weight, attn.v.bias,
            attn.e.weight, attn.e.bias, attn.out_e.weight, attn.out_e.bias, attn.out_n.weight, attn.out_n.bias,
            ln3.weight, ln3.bias, ln4.weight, ln4.bias)
    fused = (x1.is_cuda and x1.dtype in _lib.DTYPES and y.dtype == x1.dtype and C == 128 and x1.dim() == 3
             and all(t is not None for t in args))
    if not fused:
        out = _composite_attn_block(*args, alpha, ln3.eps, ln4.eps, need_edge)
    elif (not in_second_order_forward() and attn_half_supported(x1.dtype, x1.shape[1], C) and _fused_attn_half_enabled()
          and tuple(y.shape) == (x1.shape[0], x1.shape[1], x1.shape[1], C)):
        out = _AttnBlockFused.apply(*args, alpha, ln3.eps, ln4.eps, need_edge)
        return (out[0], out[1]) if need_edge else (out, None)
    else:
        prev = (None,) * 5 if y_ln is None else (y_ln.pre, y_ln.mean, y_ln.rstd, y_ln.gamma, y_ln.beta)
        out = _AttnBlock.apply(*args, alpha, ln3.eps, ln4.eps, need_edge, *prev)
        return (out[0], out[1]) if need_edge else (out[0], None)
    return out if need_edge else (out, None)


# --------------------------------------------------------------------------
# edge embedding MLP + symmetrisation (reference models.py:57-61,92-94 / 159-163,197-199)
# --------------------------------------------------------------------------
_ACT_IDS = {"relu": 0, "leaky": 1, "sigmoid": 2, "tanh": 3}
_ACT_FNS = {"relu": torch.relu, "leaky": lambda t: torch.nn.functional.leaky_relu(t, 0.01),
            "sigmoid": torch.sigmoid, "tanh": torch.tanh}
_embed_pack_cache = {}


def _embed_packed_w2(w2, dgrad: bool = False):
    key = (id(w2), dgrad)
    hit = _embed_pack_cache.get(key)
    if (hit is not None and hit[0]() is w2 and hit[1] == w2._version and hit[3] == w2.data_ptr()
            and hit[4] == _weights_epoch):
        return hit[2]
    lib = _lib.load()
    n_floats = lib.dg_embed_sym_dgrad_packed_floats() if dgrad else lib.dg_embed_sym_packed_floats()
    packed = torch.empty(int(n_floats), dtype=torch.float32, device=w2.device)
    wd = _c(w2.detach())
    with _dev(w2):
        pack = lib.dg_embed_sym_pack_dgrad if dgrad else lib.dg_embed_sym_pack
        _lib.check(pack(_lib.ptr(wd), _lib.ptr(packed), _lib.stream_of(w2)), "dg_embed_sym_pack")
    _embed_pack_cache[key] = (weakref.ref(w2), w2._version, packed, w2.data_ptr(), _weights_epoch)
    return packed


def _composite_embed_sym(a, w1, b1, w2, b2, act):
    f = _ACT_FNS[act]
    h = f(linear(a, w1, b1))
    e = f(linear(h, w2, b2))
    return (e + e.permute(0, 2, 1, 3)) / 2


class OutSlot:
    """A destination buffer handed to an embedding Function as a plain Python object (autograd never sees it): the
    kernel writes into ``tensor`` and the Function returns a fresh alias of it.  Used to let the parts of a batch (a
    one-hot real half, a dense generated half) land in ONE [sum B, N, N, C] buffer without a concatenation copy."""
    __slots__ = ("tensor",)

    def __init__(self, tensor):
        self.tensor = tensor


def _take_slot(slot, shape, dtype, device):
    if slot is None:
        return torch.empty(*shape, dtype=dtype, device=device)
    t = slot.tensor
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device or not t.is_contiguous():
        raise RuntimeError("OutSlot does not match the embedding's output")
    return t.view(shape)          # a fresh alias (forward runs with grad mode off)


class _JoinParts(Function):
    """The buffer whose dim-0 slices were filled by ``parts`` as ONE tensor of the graph: forward returns an alias of the
    buffer (no copy), backward hands each part its slice of the gradient (views)."""

    @staticmethod
    def forward(ctx, slot, *parts):
        ctx.sizes = [p.shape[0] for p in parts]
        off = 0
        for p_ in parts:
            if p_.data_ptr() != slot.tensor[off:off + p_.shape[0]].data_ptr():
                raise RuntimeError("join_parts: a part does not live in its slice of the buffer")
            off += p_.shape[0]
        return slot.tensor.view(slot.tensor.shape)

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for n in ctx.sizes:
            outs.append(g[off:off + n])
            off += n
        return (None, *outs)


def join_parts(slot, parts):
    return _JoinParts.apply(slot, *parts)


class _EmbedSym(Function):
    @staticmethod
    def forward(ctx, a, w1, b1, w2, b2, act, out_dtype, slot=None):
        a = _c(a)
        B, N, _, E = a.shape
        H, C = w1.shape[0], w2.shape[0]
        lib = _lib.load()
        out = _take_slot(slot, (B, N, N, C), out_dtype, a.device)
        with _dev(a):
            _lib.check(lib.dg_embed_sym_fwd(_lib.fptr(a), _lib.fptr(_c(w1)), _lib.fptr(_c(b1)), _lib.fptr(_embed_packed_w2(w2)),
                                            _lib.fptr(_c(b2)), _lib.ptr(out), B, N, E, H, C, _ACT_IDS[act],
                                            _lib.dt(out), _lib.stream_of(a)), "dg_embed_sym_fwd")
        _account("embed_sym", B * N * N * (4 * E + out.element_size() * C), 2 * B * N * N * (E * H + H * C))
        ctx.save_for_backward(a, w1, b1, w2, b2)
        ctx.act = act
        ctx.out_dtype = out_dtype
        return out

    @staticmethod
    def backward(ctx, g):
        a, w1, b1, w2, b2 = ctx.saved_tensors
        act = ctx.act
        if torch.is_grad_enabled():
            odt = ctx.out_dtype
            if act in _PIECEWISE_LINEAR:       # native second order (gradient penalty)
                outs = _EmbedSymBwd.apply(a, w1, b1, w2, b2, g, act, odt, ctx.needs_input_grad[0],
                                          ctx.needs_input_grad[1] and not _inputs_only())
                return tuple(outs) + (None, None, None)
            return _double_backward_fallback(lambda *t: _composite_embed_sym(*t, act).to(odt), (a, w1, b1, w2, b2), g) + (None, None, None)
        return _embed_bwd_launch(a, w1, b1, w2, b2, g, act, ctx.out_dtype, ctx.needs_input_grad[0],
                                 ctx.needs_input_grad[1] and not _inputs_only()) + (None, None, None)


_PIECEWISE_LINEAR = ("relu", "leaky")


def _embed_bwd_launch(a, w1, b1, w2, b2, g, act, out_dtype, need_da, need_w):
    B, N, _, E = a.shape
    H, C = w1.shape[0], w2.shape[0]
    lib = _lib.load()
    g = _c(g if g.dtype == out_dtype else g.to(out_dtype))
    da = torch.empty_like(a) if need_da else None
    dw1, db1, dw2, db2 = (torch.empty_like(t) for t in (w1, b1, w2, b2))
    if (out_dtype == torch.bfloat16 and act in _PIECEWISE_LINEAR and E <= 8 and N <= 48
            and options.embed_bf16 == "fast"):
        # bf16 gradients, relu / leaky: row-block streaming kernel (csrc/embed_bf16.hip)
        need = int(lib.dg_embed_sym_bwd_bf16_workspace_bytes(B, N))
        with _dev(a):
            ws = _scratch(a, need, "embed16")
            _lib.check(lib.dg_embed_sym_bwd_bf16(_lib.fptr(a), _lib.fptr(_c(w1)), _lib.fptr(_c(b1)), _lib.fptr(_c(w2)),
                                                 _lib.fptr(_c(b2)), _lib.ptr(g), _lib.ptr(da), _lib.ptr(dw1), _lib.ptr(db1),
                                                 _lib.ptr(dw2), _lib.ptr(db2), ws.data_ptr(), ws.numel(), B, N, E, H, C,
                                                 _ACT_IDS[act], _lib.stream_of(a)), "dg_embed_sym_bwd_bf16")
        _account("embed_sym", B * N * N * (4 * E * (2 if da is not None else 1) + 2 * g.element_size() * C),
                 2 * B * N * N * (E * H + H * C) * 3)
        if not need_w:
            dw1 = db1 = dw2 = db2 = None
        return da, dw1, db1, dw2, db2
    need = int(lib.dg_embed_sym_workspace_bytes(B, N))
    with _dev(a):
        ws = _scratch(a, need, "embed")
        _lib.check(lib.dg_embed_sym_bwd(_lib.fptr(a), _lib.fptr(_c(w1)), _lib.fptr(_c(b1)),
                                        _lib.fptr(_embed_packed_w2(w2)), _lib.fptr(_embed_packed_w2(w2, True)),
                                        _lib.fptr(_c(b2)), _lib.ptr(g), _lib.ptr(da), _lib.ptr(dw1), _lib.ptr(db1),
                                        _lib.ptr(dw2), _lib.ptr(db2), ws.data_ptr(), ws.numel(), B, N, E, H, C,
                                        _ACT_IDS[act], _lib.dt(g), _lib.stream_of(a)), "dg_embed_sym_bwd")
    _account("embed_sym", B * N * N * (4 * E * (2 if da is not None else 1) + g.element_size() * C),
             2 * B * N * N * (E * H + H * C) * 3)
    if not need_w:
        dw1 = db1 = dw2 = db2 = None
    return da, dw1, db1, dw2, db2


class _EmbedSymBwd(Function):
    """First backward of ``_EmbedSym`` as a differentiable node (piecewise-linear activations): its own backward is
    ``dg_embed_sym_bwd2`` -- the gradient penalty differentiates d out / d a (reference loss.py:32-39).  Only the
    adjoint of ``da`` is propagated; adjoints of the parameter gradients would need the composite graph."""

    @staticmethod
    def forward(ctx, a, w1, b1, w2, b2, g, act, out_dtype, need_da, need_w):
        ctx.save_for_backward(a, w1, b1, w2, b2, g)
        ctx.act, ctx.out_dtype = act, out_dtype
        outs = _embed_bwd_launch(a.detach(), w1.detach(), b1.detach(), w2.detach(), b2.detach(), g.detach(), act,
                                 out_dtype, need_da, need_w)
        ctx.mark_non_differentiable(*[o for o in outs[1:] if o is not None])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, t_da, t_dw1, t_db1, t_dw2, t_db2):
        a, w1, b1, w2, b2, g = ctx.saved_tensors
        if t_da is None:
            return (None,) * 10
        B, N, _, E = a.shape
        H, C = w1.shape[0], w2.shape[0]
        lib = _lib.load()
        g = _c(g if g.dtype == ctx.out_dtype else g.to(ctx.out_dtype))
        t = _c(t_da.float())
        gg = torch.empty_like(g)
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        need = int(lib.dg_embed_sym_workspace_bytes(B, N))
        with _dev(a):
            ws = _scratch(a, need, "embed")
            _lib.check(lib.dg_embed_sym_bwd2(_lib.fptr(_c(a)), _lib.fptr(_c(w1)), _lib.fptr(_c(b1)),
                                             _lib.fptr(_embed_packed_w2(w2)), _lib.fptr(_embed_packed_w2(w2, True)),
                                             _lib.fptr(_c(b2)), _lib.ptr(g), _lib.fptr(t), _lib.ptr(gg), _lib.ptr(gw1),
                                             _lib.ptr(gw2), ws.data_ptr(), ws.numel(), B, N, E, H, C,
                                             _ACT_IDS[ctx.act], _lib.dt(g), _lib.stream_of(a)), "dg_embed_sym_bwd2")
        _account("embed_sym", B * N * N * (8 * E + 2 * g.element_size() * C), 2 * B * N * N * (E * H + H * C) * 4)
        if _inputs_only() or not ctx.needs_input_grad[1]:
            gw1 = gw2 = None
        return None, gw1, None, gw2, None, gg, None, None, None, None


def embed_sym(a, w1, b1, w2, b2, act: str, out_dtype=torch.float32, slot=None):
    """(f(a) + f(a)^T(i<->j)) / 2 with f = act(W2 act(W1 a + b1) + b2): the edge embedding MLP and the
    symmetrisation of Generator / Discriminator in one kernel per direction (hidden 64, dim 128).
    The input graph ``a`` is float32; the [B,N,N,dim] result is stored as ``out_dtype``."""
    ok = (a.is_cuda and a.dtype == torch.float32 and a.dim() == 4 and a.shape[1] == a.shape[2] and act in _ACT_IDS
          and a.shape[-1] <= 16 and tuple(w1.shape) == (64, a.shape[-1]) and tuple(w2.shape) == (128, 64)
          and b1 is not None and b2 is not None)
    if not ok or (in_second_order_forward() and act not in _PIECEWISE_LINEAR):
        out = _composite_embed_sym(a, w1, b1, w2, b2, act).to(out_dtype)
        if slot is not None:
            slot.tensor = None          # the caller falls back to a concatenation
        return out
    return _EmbedSym.apply(a, w1, b1, w2, b2, act, out_dtype, slot)


# --------------------------------------------------------------------------
# one-hot input graphs: the embedding MLP collapses to an E-row table
# (reference src/data/utils.py:15-23 + models.py:57-61,92-94)
# --------------------------------------------------------------------------
def as_one_hot(a, labels=None):
    """Declare (after checking it) that the edge tensor ``a`` [B,N,N,E] is one-hot over its last dim -- true for
    every adjacency the reference's ``load_molecules`` / ``label2onehot`` produces (generator input, the
    discriminator's real batch), false for generated / interpolated tensors.  The int32 labels are attached to the
    tensor object; Generator / Discriminator then evaluate the edge-embedding MLP on the E distinct rows only
    (``dg_onehot_embed_fwd/bwd``).  The check costs one device->host read per NEW tensor object (the result is
    cached on it together with the tensor's version counter: an in-place write invalidates it), so a resident batch
    is checked once.  Returns ``a``."""
    if not (torch.is_tensor(a) and a.is_cuda and a.dim() == 4):
        return a
    if getattr(a, "_dg_labels", None) is not None and getattr(a, "_dg_labels_version", None) == a._version:
        return a                      # same object, not written since the check (in-place updates bump _version)
    a._dg_labels_version = a._version
    if a.requires_grad or a.dtype != torch.float32:
        a._dg_labels = False
        return a
    with torch.no_grad():
        if labels is None:
            labels = a.argmax(-1).to(torch.int32)
        ok = ((a.amax(-1) == 1) & (a.sum(-1) == 1) & (a.amin(-1) == 0)).all() if a.shape[-1] > 1 else (a == 1).all()
    a._dg_labels = labels.contiguous() if bool(ok) else False     # the only host sync: once per tensor object
    return a


def attach_one_hot_labels(a, labels):
    """Declare WITHOUT checking that ``a`` [B,N,N,E] is one-hot with the given int32 ``labels`` [B,N,N] -- for producers
    that build ``a`` from the labels (``data.dense_one_hot_adjacency``: reference utils.py:15-23,130-137) or refresh both
    together (``GraphedGANStep``).  No device->host sync.  Returns ``a``."""
    if labels.dtype != torch.int32 or tuple(labels.shape) != tuple(a.shape[:-1]) or labels.device != a.device:
        raise ValueError("labels must be an int32 tensor on a's device with a's shape minus the last dim")
    a._dg_labels = labels if labels.is_contiguous() else labels.contiguous()
    a._dg_labels_version = a._version
    return a


def one_hot_labels(a):
    """The int32 labels attached by ``as_one_hot`` (None for tensors that are not declared one-hot)."""
    lab = getattr(a, "_dg_labels", None)
    if not torch.is_tensor(lab) or getattr(a, "_dg_labels_version", None) != a._version:
        return None                   # never declared, not one-hot, or written in place since the check
    return lab


class _OneHotEmbed(Function):
    @staticmethod
    def forward(ctx, labels, table, out_dtype, slot=None):
        B, N = labels.shape[0], labels.shape[1]
        E, C = table.shape
        lib = _lib.load()
        table = _c(table)
        out = _take_slot(slot, (B, N, N, C), out_dtype, labels.device)
        with _dev(labels):
            _lib.check(lib.dg_onehot_embed_fwd(labels.data_ptr(), _lib.fptr(table), _lib.ptr(out), B, N, E, C, _lib.dt(out),
                                               _lib.stream_of(out)), "dg_onehot_embed_fwd")
        _account("embed_sym", B * N * N * (8 + out.element_size() * C))
        ctx.save_for_backward(labels)
        ctx.shape = (E, C, out_dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (labels,) = ctx.saved_tensors
        E, C, out_dtype = ctx.shape
        B, N = labels.shape[0], labels.shape[1]
        lib = _lib.load()
        g = _c(g if g.dtype == out_dtype else g.to(out_dtype))
        dtable = torch.empty(E, C, dtype=torch.float32, device=g.device)
        need = int(lib.dg_onehot_embed_workspace_bytes(E, C))
        with _dev(g):
            ws = _scratch(g, need, "onehot")
            _lib.check(lib.dg_onehot_embed_bwd(labels.data_ptr(), _lib.ptr(g), _lib.ptr(dtable), ws.data_ptr(), ws.numel(),
                                               B, N, E, C, _lib.dt(g), _lib.stream_of(g)), "dg_onehot_embed_bwd")
        _account("embed_sym", B * N * N * (8 + g.element_size() * C))
        return None, dtable, None, None


def embed_sym_onehot(labels, w1, b1, w2, b2, act: str, out_dtype=torch.float32, slot=None):
    """``embed_sym`` for a one-hot input given by its labels [B,N,N]: the MLP runs on the E unit vectors (plain torch
    ops on [E,64] / [E,128] tensors, differentiated by autograd), the [B,N,N,dim] result is a symmetrised gather."""
    f = _ACT_FNS[act]
    table = f(torch.nn.functional.linear(f(w1.t() + b1), w2, b2))      # [E, dim]: row c = f(one_hot(c))
    return _OneHotEmbed.apply(labels, table, out_dtype, slot)
