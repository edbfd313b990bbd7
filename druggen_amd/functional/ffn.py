"""Feed-forward half of an Encoder_Block (reference src/model/layers.py:41-54,191-192): fc1 + ReLU + fc2 + residual + LayerNorm as
one autograd node -- float32 (fused forward kernel or two row GEMMs, node + edge halves riding in one launch) and bf16."""
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
from .dense import *      # noqa: F401,F403
from .heads import *      # noqa: F401,F403


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
            and hit[4] == (w1.data_ptr(), w2.data_ptr()) and hit[5] == _weights_epoch[0]):
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
                                (w1.data_ptr(), w2.data_ptr()), _weights_epoch[0])
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
            and hit[4] == (w1.data_ptr(), w2.data_ptr()) and hit[5] == _weights_epoch[0]):
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
                            (w1.data_ptr(), w2.data_ptr()), _weights_epoch[0])
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


__all__ = [_n for _n in dir() if not _n.startswith("__")]
