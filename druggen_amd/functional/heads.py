"""The small layers at both ends of the networks: the Generator's readouts, the node embedding chain and the Discriminator's head
(reference src/model/models.py:63-66,95-101,165-171,200-207)."""
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


__all__ = [_n for _n in dir() if not _n.startswith("__")]
