"""Proves the closed forms in tests/kernel_math.py (the spec the HIP kernels
implement) against torch autograd, first and second order, in float64."""
import torch

import kernel_math as km

torch.manual_seed(0)
DT = torch.float64


def _rand(*shape):
    return torch.randn(*shape, dtype=DT)


def test_attn_core_first_and_second_order():
    B, N, C, alpha = 2, 5, 6, 0.37
    q, k, v = (_rand(B, N, C).requires_grad_() for _ in range(3))
    e = _rand(B, N, N, C).requires_grad_()
    ws, wo = _rand(B, N, N, C).requires_grad_(), _rand(B, N, C).requires_grad_()
    s, o = km.attn_core_fwd(q, k, v, e, alpha)
    grads = torch.autograd.grad([s, o], [q, k, v, e], [ws, wo], create_graph=True)
    mine = km.attn_core_bwd(q, k, v, e, ws, wo, alpha)
    for a, b in zip(grads, mine):
        assert torch.allclose(a, b, rtol=1e-11, atol=1e-12)
    t = [_rand(*g.shape) for g in grads]
    phi = sum((g * tt).sum() for g, tt in zip(grads, t))
    second = torch.autograd.grad(phi, [q, k, v, e, ws, wo])
    mine2 = km.attn_core_bwd2(q, k, v, e, ws, wo, *t, alpha)
    for name, a, b in zip("q k v e ws wo".split(), second, mine2):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-11), name


def test_layernorm_first_and_second_order():
    R, C = 7, 10
    z = _rand(3, R, C).requires_grad_()
    gamma, beta = (1 + 0.1 * _rand(C)).requires_grad_(), _rand(C).requires_grad_()
    dy = _rand(3, R, C).requires_grad_()
    y = torch.nn.functional.layer_norm(z, (C,), gamma, beta, 1e-5)
    y2, mu, rstd = km.ln_fwd(z, gamma, beta)
    assert torch.allclose(y, y2, rtol=1e-12, atol=1e-12)
    dz, dg, db = torch.autograd.grad(y, [z, gamma, beta], dy, create_graph=True)
    mz, mg, mb = km.ln_bwd(z, gamma, mu, rstd, dy)
    assert torch.allclose(dz, mz, rtol=1e-11, atol=1e-12)
    assert torch.allclose(dg, mg, rtol=1e-11, atol=1e-12)
    assert torch.allclose(db, mb, rtol=1e-11, atol=1e-12)
    tz = _rand(3, R, C)
    second = torch.autograd.grad((dz * tz).sum(), [z, gamma, dy])
    mine = km.ln_bwd2(z, gamma, mu, rstd, dy, tz)
    for name, a, b in zip("z gamma dy".split(), second, mine):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-11), name
