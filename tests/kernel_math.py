"""Closed forms the HIP kernels implement, written with plain torch ops.

These are the *specification* of ``druggen_amd/csrc``: each function states the
math of one C-ABI entry point.  ``tests/test_kernel_math.py`` proves them
against torch autograd (first and second order) in float64 on CPU; the GPU
tests then compare the kernels with them.

Notation (reference ``src/model/layers.py:119-134``): per channel c,
    s_ij = alpha q_i k_j g(e_ij),  g(e) = e^2 + e
    p_ij = softmax_j s_ij,         o_i  = sum_j p_ij v_j
"""
from __future__ import annotations

import torch


def attn_core_fwd(q, k, v, e, alpha):
    g = e * e + e
    s = alpha * q.unsqueeze(2) * k.unsqueeze(1) * g
    p = torch.softmax(s, dim=2)
    o = (p * v.unsqueeze(1)).sum(2)
    return s, o


def attn_core_bwd(q, k, v, e, ws, wo, alpha):
    """(ws, wo) = upstream grads of (s, o) -> (dq, dk, dv, de)."""
    g, g1 = e * e + e, 2 * e + 1
    qi, kj = q.unsqueeze(2), k.unsqueeze(1)
    s = alpha * qi * kj * g
    p = torch.softmax(s, dim=2)
    a = wo.unsqueeze(2) * v.unsqueeze(1)
    abar = (p * a).sum(2, keepdim=True)
    ds = ws + p * (a - abar)
    dv = (p * wo.unsqueeze(2)).sum(1)
    dq = (ds * alpha * kj * g).sum(2)
    dk = (ds * alpha * qi * g).sum(1)
    de = ds * alpha * qi * kj * g1
    return dq, dk, dv, de


def attn_core_bwd2(q, k, v, e, ws, wo, tq, tk, tv, te, alpha):
    """Backward of ``attn_core_bwd``: (tq,tk,tv,te) are the adjoints of its
    outputs (dq,dk,dv,de).  Returns adjoints of its inputs in the order
    (q, k, v, e, ws, wo)."""
    g, g1 = e * e + e, 2 * e + 1
    qi, kj, vj = q.unsqueeze(2), k.unsqueeze(1), v.unsqueeze(1)
    tqi, tkj, tvj = tq.unsqueeze(2), tk.unsqueeze(1), tv.unsqueeze(1)
    woi = wo.unsqueeze(2)
    s = alpha * qi * kj * g
    p = torch.softmax(s, dim=2)
    a = woi * vj
    abar = (p * a).sum(2, keepdim=True)
    ds = ws + p * (a - abar)
    # tangent of the forward along t
    sdot = alpha * (tqi * kj * g + qi * tkj * g + qi * kj * g1 * te)
    m = (p * sdot).sum(2, keepdim=True)
    pdot = p * (sdot - m)
    odot = (pdot * vj + p * tvj).sum(2)
    # gradient of <sdot, ws> + <odot, wo> w.r.t. the primal inputs
    b = woi * tvj
    pbar = sdot * (a - abar) - m * a + b
    sbar = p * (pbar - (p * pbar).sum(2, keepdim=True))
    qbar = (sbar * alpha * kj * g + ds * alpha * (tkj * g + kj * g1 * te)).sum(2)
    kbar = (sbar * alpha * qi * g + ds * alpha * (tqi * g + qi * g1 * te)).sum(1)
    ebar = sbar * alpha * qi * kj * g1 + ds * alpha * (tqi * kj * g1 + qi * tkj * g1 + 2 * qi * kj * te)
    vbar = (pdot * woi).sum(1)
    return qbar, kbar, vbar, ebar, sdot, odot


# ---------------------------------------------------------------------------
# residual + LayerNorm:  y = LN(a + r) * gamma + beta   (layers.py:185-192)
# ---------------------------------------------------------------------------
def ln_fwd(z, gamma, beta, eps=1e-5):
    mu = z.mean(-1, keepdim=True)
    var = ((z - mu) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    xhat = (z - mu) * rstd
    return xhat * gamma + beta, mu, rstd


def ln_bwd(z, gamma, mu, rstd, dy):
    """-> (dz, dgamma, dbeta)."""
    xhat = (z - mu) * rstd
    gdy = dy * gamma
    c1 = gdy.mean(-1, keepdim=True)
    c2 = (gdy * xhat).mean(-1, keepdim=True)
    dz = rstd * (gdy - c1 - xhat * c2)
    red = tuple(range(z.dim() - 1))
    return dz, (dy * xhat).sum(red), dy.sum(red)


def ln_bwd2(z, gamma, mu, rstd, dy, tz):
    """Backward of ``ln_bwd`` restricted to the adjoint ``tz`` of dz (the only
    non-zero one inside the gradient penalty, where the first backward is taken
    w.r.t. inputs only).  Returns adjoints of (z, gamma, dy)."""
    xhat = (z - mu) * rstd
    gdy = dy * gamma
    c2 = (gdy * xhat).mean(-1, keepdim=True)
    # tangent of xhat along tz
    t1 = tz.mean(-1, keepdim=True)
    t2 = (tz * xhat).mean(-1, keepdim=True)
    xdot = rstd * (tz - t1 - xhat * t2)            # = d xhat[tz]
    dy_bar = gamma * xdot                          # adjoint of dy
    gamma_bar = (dy * xdot).sum(tuple(range(z.dim() - 1)))
    # adjoint of z:  d/dz < xdot(z), gdy >
    u = gdy
    u1 = u.mean(-1, keepdim=True)
    # xdot = rstd * (tz - t1 - xhat * t2);  d(rstd) = -rstd^2 * mean(xhat * dz') ...
    # closed form (derived in DESIGN.md, verified against autograd):
    w = rstd * (u - u1 - xhat * c2)                # = dz of the first order with dy
    s1 = (xdot * u).mean(-1, keepdim=True)
    z_bar = -rstd * (xhat * s1 + xdot * c2 + w * t2)
    return z_bar, gamma_bar, dy_bar
