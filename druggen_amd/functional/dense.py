"""nn.Linear on the library: weight gradients (dg_linear_wgrad*), packed weights and their cache, the row GEMMs with fused prologue /
epilogue (dg_row_gemm*), three Linears per launch (q / k / v), LayerNorm backward inside a neighbouring GEMM, Linear + LayerNorm."""
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
from ._runtime import *      # noqa: F401,F403
from .layernorm import *      # noqa: F401,F403


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


# --------------------------------------------------------------------------
# fp32-MFMA row GEMM with fused prologue / epilogue (dg_row_gemm)
# --------------------------------------------------------------------------
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


def packed_weight(w, mode: int, dtype=torch.float32):
    """MFMA-fragment-ordered copy of an nn.Linear weight (mode 0: forward, 1: input
    gradient) for activations of ``dtype``, cached per (storage, version): re-packed only after an
    optimizer step."""
    w = _canon(w)
    key = (id(w), mode, dtype)
    hit = _pack_cache.get(key)
    if (hit is not None and hit[0]() is w and hit[1] == w._version and hit[3] == w.data_ptr()
            and hit[4] == _weights_epoch[0]):
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
    _pack_cache[key] = (weakref.ref(w), w._version, packed, w.data_ptr(), _weights_epoch[0])
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
            and hit[3] == tuple(w.data_ptr() for w in ws) and hit[4] == _weights_epoch[0]):
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
                         tuple(w.data_ptr() for w in ws), _weights_epoch[0])
    return packed


def lin3_supported(x2, ws) -> bool:
    """Three Linear(128,128) per launch (dg_row_gemm_lin3 / _sum3, dg_linear_wgrad3): float32 rows on the fp16 hi + lo
    kernels."""
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
            cache[key] = (weakref.ref(ws[0]), ws[0]._version, packed, ws[0].data_ptr(), _weights_epoch[0])
        else:
            cache[key] = (tuple(weakref.ref(w) for w in ws), tuple(w._version for w in ws), packed,
                          tuple(w.data_ptr() for w in ws), _weights_epoch[0])
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


__all__ = [_n for _n in dir() if not _n.startswith("__")]
