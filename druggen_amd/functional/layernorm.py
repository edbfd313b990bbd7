"""Residual + LayerNorm (reference src/model/layers.py:185-192): forward, backward and the backward's backward."""
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


__all__ = [_n for _n in dir() if not _n.startswith("__")]
