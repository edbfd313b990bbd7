"""State every autograd node family shares: per-kernel traffic accounting for bench.py, the storage modes of the feed-forward's
hidden tensors, scratch workspaces, the activation dtype, the pass flags (inputs-only backward, second-order forward) and the
thread-local scopes of the C library (reduce batches, riding launches)."""
from __future__ import annotations

import contextlib
import ctypes
import threading
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from ..options import options


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


def _alias_outputs_enabled() -> bool:
    return options.penalty_wgrad == "joined"


_pack_cache = {}            # packed weights of the row GEMMs (dense.packed_weight)
_embed_pack_cache = {}      # packed second layer of the edge embedding (embed._embed_packed_w2)
_weights_epoch = [0]        # a one-element list: every module that caches packed weights reads the same counter


def bump_weights_epoch() -> None:
    """Invalidate every packed weight (both caches share this epoch).  Called by writers that change
    parameters behind autograd's back: ``FlatAdamW.step`` (raw kernel on the flat buffer) and
    ``GraphedGANStep`` (before capture, so that the first use after each optimizer step records its
    pack kernel into the graph, and after every replay, which updates weights without touching
    ``tensor._version``)."""
    _weights_epoch[0] += 1
    if len(_pack_cache) + len(_embed_pack_cache) > 8192:
        with _cache_lock:
            for cache in (_pack_cache, _embed_pack_cache):
                for k in [k for k, v in list(cache.items()) if v[0]() is None]:
                    cache.pop(k, None)


__all__ = [_n for _n in dir() if not _n.startswith("__")]
