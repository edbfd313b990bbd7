"""GPU parity of the C-ABI kernels against the closed forms (tests/kernel_math.py,
themselves proven against autograd) evaluated in float64 on the CPU."""
import math

import pytest
import torch

import kernel_math as km

pytestmark = pytest.mark.gpu

TOL = 2e-5     # fp32 kernels vs fp64 closed form, relative L2 (north_star bar: 1e-3)


def _rel(got, want):
    want = want.double()
    den = want.norm().item()
    return (got.double().cpu() - want).norm().item() / (den if den > 0 else 1.0)


def _lib():
    from druggen_amd import _lib
    return _lib


def _gen(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


ATTN_SHAPES = [(2, 5, 8), (3, 6, 16), (2, 7, 12), (2, 9, 128), (1, 17, 48), (2, 24, 32), (2, 45, 128),
               (1, 90, 128), (1, 33, 256), (3, 1, 16), (1, 96, 64)]


@pytest.mark.parametrize("B,N,C", ATTN_SHAPES)
def test_attn_core_forward_backward_second_order(B, N, C):
    from druggen_amd import functional as dgf
    alpha = 1.0 / math.sqrt(C // 4 if C >= 4 else 1)
    q, k, v = (_gen((B, N, C), s) for s in (1, 2, 3))
    e = _gen((B, N, N, C), 4, 0.8)
    ws, wo = _gen((B, N, N, C), 5), _gen((B, N, C), 6)
    t = [_gen((B, N, C), 7), _gen((B, N, C), 8), _gen((B, N, C), 9), _gen((B, N, N, C), 10)]
    s_ref, o_ref = km.attn_core_fwd(q, k, v, e, alpha)
    g_ref = km.attn_core_bwd(q, k, v, e, ws, wo, alpha)
    h_ref = km.attn_core_bwd2(q, k, v, e, ws, wo, *t, alpha)

    dev = "cuda"
    f = lambda x: x.float().to(dev).requires_grad_(True)
    qd, kd, vd, ed, wsd, wod = map(f, (q, k, v, e, ws, wo))
    s, o = dgf.attn_core(qd, kd, vd, ed, alpha)
    assert _rel(s, s_ref) < TOL and _rel(o, o_ref) < TOL
    grads = torch.autograd.grad([s, o], [qd, kd, vd, ed], [wsd, wod], create_graph=True)
    for name, got, want in zip("dq dk dv de".split(), grads, g_ref):
        assert _rel(got, want) < TOL, name
    phi = sum((g * tt.float().to(dev)).sum() for g, tt in zip(grads, t))
    second = torch.autograd.grad(phi, [qd, kd, vd, ed, wsd, wod])
    for name, got, want in zip("gq gk gv ge gws gwo".split(), second, h_ref):
        assert _rel(got, want) < 5 * TOL, name


def test_attn_core_without_score_output_and_null_ws():
    """Discriminator's last block: s is neither written nor differentiated."""
    from druggen_amd import functional as dgf
    B, N, C, alpha = 2, 11, 64, 0.25
    q, k, v, e = _gen((B, N, C), 1), _gen((B, N, C), 2), _gen((B, N, C), 3), _gen((B, N, N, C), 4, 0.7)
    wo = _gen((B, N, C), 5)
    _, o_ref = km.attn_core_fwd(q, k, v, e, alpha)
    g_ref = km.attn_core_bwd(q, k, v, e, torch.zeros_like(e), wo, alpha)
    f = lambda x: x.float().cuda().requires_grad_(True)
    qd, kd, vd, ed = map(f, (q, k, v, e))
    s, o = dgf.attn_core(qd, kd, vd, ed, alpha, need_s=False)
    assert s is None and _rel(o, o_ref) < TOL
    grads = torch.autograd.grad(o, [qd, kd, vd, ed], wo.float().cuda(), create_graph=True)
    for got, want in zip(grads, g_ref):
        assert _rel(got, want) < TOL
    # second order with ws == NULL
    t = [_gen((B, N, C), 7), _gen((B, N, C), 8), _gen((B, N, C), 9), _gen((B, N, N, C), 10)]
    h_ref = km.attn_core_bwd2(q, k, v, e, torch.zeros_like(e), wo, *t, alpha)
    phi = sum((g * tt.float().cuda()).sum() for g, tt in zip(grads, t))
    second = torch.autograd.grad(phi, [qd, kd, vd, ed])
    for got, want in zip(second, h_ref[:4]):
        assert _rel(got, want) < 5 * TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,C", [(2, 9, 128), (3, 45, 128), (1, 90, 128), (2, 7, 32)])
def test_attn_core_backward_adds_an_outside_adjoint_of_e(B, N, C, dtype):
    """dg_attn_core_bwd_add: the adjoint that the gradient penalty's second order hands to the first-order pass
    (loss.py:32-47) joins de inside the kernel; dq / dk / dv are untouched, de = de(plain launch) + add_e."""
    from druggen_amd import functional as dgf
    f = lambda shape, s, sc=1.0: (_gen(shape, s) * sc).to(dtype).cuda()
    q, k, v, wo = f((B, N, C), 1), f((B, N, C), 2), f((B, N, C), 3), f((B, N, C), 6)
    e, ws, ae = f((B, N, N, C), 4, 0.8), f((B, N, N, C), 5), f((B, N, N, C), 11, 0.5)
    plain = dgf._attn_bwd_launch(q, k, v, e, ws, wo, 0.25)
    fused = dgf._attn_bwd_launch(q, k, v, e, ws, wo, 0.25, add_e=ae)
    for a, b in zip(plain[:3], fused[:3]):
        assert torch.equal(a, b)
    want = plain[3].double() + ae.double()
    tol = 1e-6 if dtype == torch.float32 else 4e-3      # bf16: the plain launch rounded de once more before the add
    assert _rel(fused[3], want.cpu()) < tol
    # without the score gradient (Discriminator's last block)
    p2 = dgf._attn_bwd_launch(q, k, v, e, None, wo, 0.25)
    f2 = dgf._attn_bwd_launch(q, k, v, e, None, wo, 0.25, add_e=ae)
    assert _rel(f2[3], (p2[3].double() + ae.double()).cpu()) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,N", [((3, 9, 9, 128), 5), ((2, 45, 128), 13), ((1, 7, 7, 128), 1), ((2, 20, 20, 128), 16), ((1, 1, 128), 10)])
def test_skinny_readout_matches_linear_on_float32_logits(shape, N, dtype):
    """dgf.readout (Generator.readout_e / readout_n, reference models.py:67-68,100-101): float32 logits from float32 or
    bf16 activations in one streaming kernel; input / weight / bias gradients against autograd of
    F.linear(x.float(), w, b) on the same (already rounded) activations."""
    from druggen_amd import functional as dgf
    x0 = _gen(shape, 1).to(dtype)
    w0, b0 = _gen((N, 128), 2) * 0.2, _gen((N,), 3)
    dy = _gen(shape[:-1] + (N,), 4)
    xr, wr, br = x0.double().requires_grad_(True), w0.double().requires_grad_(True), b0.double().requires_grad_(True)
    y_ref = torch.nn.functional.linear(xr, wr, br)
    g_ref = torch.autograd.grad(y_ref, [xr, wr, br], dy)
    xd = x0.cuda().requires_grad_(True)
    wd, bd = w0.float().cuda().requires_grad_(True), b0.float().cuda().requires_grad_(True)
    y = dgf.readout(xd, wd, bd)
    assert y.dtype == torch.float32 and type(y.grad_fn).__name__.startswith("_Readout")
    assert _rel(y, y_ref) < TOL
    gx, gw, gb = torch.autograd.grad(y, [xd, wd, bd], dy.float().cuda())
    assert gx.dtype == dtype
    assert _rel(gx, g_ref[0]) < (TOL if dtype == torch.float32 else 4e-3)      # dx is stored in the activation dtype
    assert _rel(gw, g_ref[1]) < TOL and _rel(gb, g_ref[2]) < TOL
    # no bias, and a graph that is differentiated again (composite fallback)
    y2 = dgf.readout(xd, wd)
    assert _rel(y2, torch.nn.functional.linear(xr, wr)) < TOL
    (g1,) = torch.autograd.grad(y2, [xd], dy.float().cuda(), create_graph=True)
    (g2,) = torch.autograd.grad((g1.float() ** 2).sum(), [wd])
    assert torch.isfinite(g2).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_repack_equals_single_packs(dtype):
    """dg_row_gemm_pack_batch (one launch for all stale packs of an optimizer's parameters) writes byte-for-byte what
    the single pack launches write, for every shape / mode of the step, and leaves other parameters' packs alone."""
    from druggen_amd import functional as dgf
    torch.manual_seed(0)
    ws = [torch.randn(128, 128, device="cuda"), torch.randn(384, 128, device="cuda"), torch.randn(128, 384, device="cuda"),
          torch.randn(128, 128, device="cuda")]
    other = torch.randn(128, 128, device="cuda")
    packs = [(w, m, dgf.packed_weight(w, m, dtype)) for w in ws for m in (0, 1)]
    keep = dgf.packed_weight(other, 0, dtype)
    keep0 = keep.clone()
    for w in ws:
        w.mul_(1.5).add_(0.01)                      # an "optimizer step": versions move
    n = dgf.repack_params(ws)
    assert n == len(packs)
    for w, m, p in packs:
        again = dgf.packed_weight(w, m, dtype)
        assert again.data_ptr() == p.data_ptr()       # cache hit: refreshed in place by the batched launch
        dgf._pack_cache.pop((id(w), m, dtype))
        single = dgf.packed_weight(w, m, dtype)       # a fresh single pack of the same weight
        assert torch.equal(single, p)
    assert torch.equal(keep, keep0)


@pytest.mark.parametrize("R", [1, 63, 200, 4097, 23040])
def test_three_linears_per_launch_match_three_launches(R):
    """q / k / v of an attention block (reference layers.py:111-113) as ONE launch per direction: dg_row_gemm_lin3 (three
    outputs), dg_row_gemm_sum3 (dq Wq + dk Wk + dv Wv + residual), dg_linear_wgrad3 (stacked [384,128] weight gradient)
    against the fp64 contractions AND against the three separate launches; the stacked packs refreshed by the batched
    repack equal fresh packs."""
    from druggen_amd import functional as dgf
    ws = [(_gen((128, 128), 30 + i) * 0.2).float().cuda() for i in range(3)]
    bs = [_gen((128,), 40 + i).float().cuda() for i in range(3)]
    x = _gen((R, 128), 50).float().cuda()
    dys = [_gen((R, 128), 60 + i).float().cuda() for i in range(3)]
    res = _gen((R, 128), 70).float().cuda()
    assert dgf.lin3_supported(x, ws)
    ys = dgf.lin3(x, ws, bs)
    for y, w, b in zip(ys, ws, bs):
        assert _rel(y, x.double().cpu() @ w.double().cpu().t() + b.double().cpu()) < TOL
        one = dgf.row_gemm(x, dgf.packed_weight(w, 0), 128, 128, bias=b)
        assert _rel(y, one.double().cpu()) < 2e-6
    nb = dgf.lin3(x, ws, (None, None, None))
    assert _rel(nb[1], x.double().cpu() @ ws[1].double().cpu().t()) < TOL
    want = sum(d.double().cpu() @ w.double().cpu() for d, w in zip(dys, ws))
    assert _rel(dgf.sum3(*dys, ws), want) < TOL
    assert _rel(dgf.sum3(*dys, ws, residual=res), want + res.double().cpu()) < TOL
    dw, db = dgf._wgrad3(dys, x, True)
    for i, d in enumerate(dys):
        assert _rel(dw[128 * i:128 * (i + 1)], d.double().cpu().t() @ x.double().cpu()) < TOL
        assert _rel(db[128 * i:128 * (i + 1)], d.double().cpu().sum(0)) < TOL
        one, _ = dgf._wgrad(d, x, True)
        assert _rel(dw[128 * i:128 * (i + 1)], one.double().cpu()) < 2e-6
    dw2, _ = dgf._wgrad3(dys, x, False)
    assert torch.equal(dw, dw2)
    # an "optimizer step", then the batched repack of everything cached for these weights
    packs = [(m, dgf.packed_weight3(*ws, m)) for m in (0, 1)]
    for w in ws:
        w.mul_(1.25).add_(0.01)
    assert dgf.repack_params(ws) >= 2
    for m, p in packs:
        again = dgf.packed_weight3(*ws, m)
        assert again.data_ptr() == p.data_ptr()
        dgf._pack3_cache.pop((id(ws[0]), id(ws[1]), id(ws[2]), m))
        assert torch.equal(dgf.packed_weight3(*ws, m), p)
    assert _rel(dgf.lin3(x, ws, bs)[2], x.double().cpu() @ ws[2].double().cpu().t() + bs[2].double().cpu()) < TOL


@pytest.mark.parametrize("Rn,Re", [(45, 2025), (360, 16200), (11520, 518400), (23040, 40000)])
def test_riding_launches_equal_separate_launches(Rn, Re):
    """dg_launch_pair_begin / _end: a node-level launch of a 384-wide row GEMM or of a producer / consumer weight
    gradient waits and runs as the second problem of the next launch of the same kernel (workgroups split by rows).
    Row GEMMs are row-local: bit-identical to separate launches, ReLU bit masks included; weight gradients use another
    (fixed) number of partial sums: equal to rounding, and reproducible.  A problem nobody carries leaves at _pair_end."""
    from druggen_amd import functional as dgf
    w1 = [(_gen((384, 128), 100 + i) * 0.1).float().cuda() for i in range(2)]
    b1 = [_gen((384,), 110 + i).float().cuda() for i in range(2)]
    w2 = [(_gen((128, 384), 120 + i) * 0.1).float().cuda() for i in range(2)]
    b2 = [_gen((128,), 130 + i).float().cuda() for i in range(2)]
    g = [(_gen((128,), 140 + i) * 0.1 + 1).float().cuda() for i in range(2)]
    be = [_gen((128,), 150 + i).float().cuda() for i in range(2)]
    xs = [_gen((R, 128), 160 + i).float().cuda() for i, R in enumerate((Rn, Re))]

    def chain(paired):
        out = []
        with dgf._pair_launches(xs[0], on=paired):
            hs = [dgf.row_gemm(x, dgf.packed_weight(w1[i], 0), 128, 384, bias=b1[i], relu=True, want_relu_bits=True)
                  for i, x in enumerate(xs)]
            ys = [dgf.row_gemm(h, dgf.packed_weight(w2[i], 0), 384, 128, bias=b2[i], residual=xs[i], ln=(g[i], be[i], 1e-5),
                               want_pre=True) for i, (h, _) in enumerate(hs)]
            dh = [dgf.row_gemm(y[0], dgf.packed_weight(w2[i], 1), 128, 384, mask_bits=hs[i][1]) for i, y in enumerate(ys)]
            dx = [dgf.row_gemm(d, dgf.packed_weight(w1[i], 1), 384, 128, residual=ys[i][0]) for i, d in enumerate(dh)]
            wg = dgf._wgrad_many([(ys[0][0], hs[0][0], True), (ys[1][0], hs[1][0], True), (dh[0], xs[0], True), (dh[1], xs[1], True)])
        for i in range(2):      # (the ReLU bit words are compared through dh: their buffer is sized for every 128 -> 384 kernel)
            out.append([hs[i][0], *ys[i], dh[i], dx[i]])
        return out, wg

    sep, wsep = chain(False)
    par, wpar = chain(True)
    for a, b in zip(sep, par):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    for (dw, db), (ew, eb) in zip(wsep, wpar):
        assert _rel(ew, dw.double().cpu()) < 2e-6 and _rel(eb, db.double().cpu()) < 2e-6
    assert _rel(wpar[0][0], sep[0][1].double().cpu().t() @ sep[0][0].double().cpu()) < TOL
    assert _rel(wpar[3][0], sep[1][5].double().cpu().t() @ xs[1].double().cpu()) < TOL
    again, wagain = chain(True)
    for (dw, db), (ew, eb) in zip(wpar, wagain):
        assert torch.equal(dw, ew) and torch.equal(db, eb)
    # regions nest: the inner end leaves the waiting launch to the outer region
    with dgf._pair_launches(xs[0]):
        hn = dgf.row_gemm(xs[0], dgf.packed_weight(w1[0], 0), 128, 384, bias=b1[0])
        with dgf._pair_launches(xs[0]):
            pass
        he = dgf.row_gemm(xs[1], dgf.packed_weight(w1[1], 0), 128, 384, bias=b1[1])
    assert torch.equal(hn, dgf.row_gemm(xs[0], dgf.packed_weight(w1[0], 0), 128, 384, bias=b1[0]))
    assert torch.equal(he, dgf.row_gemm(xs[1], dgf.packed_weight(w1[1], 0), 128, 384, bias=b1[1]))
    # a rider nobody carries: launched by _pair_end
    with dgf._pair_launches(xs[0]):
        lone = dgf.row_gemm(xs[0], dgf.packed_weight(w1[0], 0), 128, 384, bias=b1[0])
    assert torch.equal(lone, dgf.row_gemm(xs[0], dgf.packed_weight(w1[0], 0), 128, 384, bias=b1[0]))
    # a rider whose 384 -> 128 epilogue differs from the next launch's: on its own, in order
    with dgf._pair_launches(xs[0]):
        plain = dgf.row_gemm(sep[0][0], dgf.packed_weight(w2[0], 0), 384, 128)
        withres = dgf.row_gemm(sep[1][0], dgf.packed_weight(w2[1], 0), 384, 128, residual=xs[1])
    assert torch.equal(plain, dgf.row_gemm(sep[0][0], dgf.packed_weight(w2[0], 0), 384, 128))
    assert torch.equal(withres, dgf.row_gemm(sep[1][0], dgf.packed_weight(w2[1], 0), 384, 128, residual=xs[1]))


@pytest.mark.parametrize("B,N", [(2, 9), (8, 45)])
def test_paired_feed_forward_nodes_match_the_two_single_nodes(B, N):
    """ffn_ln_pair (mlp / ln5 over the node rows + mlp2 / ln6 over the edge rows of an Encoder_Block, reference
    layers.py:191-192, as one autograd node) against the two ffn_ln nodes: forward and input gradients bit-identical,
    parameter gradients to rounding, and the same through a double backward (gradient-penalty pattern, loss.py:28-39)."""
    from druggen_amd import functional as dgf
    C, H = 128, 384
    mk = lambda s, shape, scale=1.0: (_gen(shape, s) * scale).float().cuda().requires_grad_(True)
    params = [[mk(200 + 10 * i, (H, C), 0.1), mk(201 + 10 * i, (H,)), mk(202 + 10 * i, (C, H), 0.1), mk(203 + 10 * i, (C,)),
               mk(204 + 10 * i, (C,)), mk(205 + 10 * i, (C,))] for i in range(2)]
    x0, y0 = mk(230, (B, N, C)), mk(231, (B, N, N, C))
    gx, gy = _gen((B, N, C), 232).float().cuda(), _gen((B, N, N, C), 233).float().cuda()

    from druggen_amd.options import options

    def run(mode, second):
        x, y = x0.detach().clone().requires_grad_(True), y0.detach().clone().requires_grad_(True)
        with options.override(ffn_pair=(mode == "on")):
            xo, yo, _ = dgf.ffn_ln_pair(x, (*params[0], 1e-5), y, (*params[1], 1e-5))
        flat = [x, y] + params[0] + params[1]
        if not second:
            return [xo, yo] + list(torch.autograd.grad([xo, yo], flat, [gx, gy]))
        with dgf.inputs_only_backward():
            dx, dy = torch.autograd.grad([xo, yo], [x, y], [gx, gy], create_graph=True)
        pen = (dx.square().sum() + dy.square().sum()) * 0.5
        return [dx, dy] + list(torch.autograd.grad(pen, flat, allow_unused=True))      # (biases / beta: no second-order term)

    for second in (False, True):
        one, two = run("off", second), run("on", second)
        for i, (a, b) in enumerate(zip(one, two)):
            if a is None or b is None:
                assert a is None and b is None, i
            elif i < 4:
                assert torch.equal(a, b), i
            else:
                assert _rel(b, a.double().cpu()) < 5e-6, i


@pytest.mark.parametrize("R", [65536, 70001, 131083])
def test_alternating_traversal_is_invisible_in_the_results(R):
    """Edge-level row GEMMs and attention launches walk the rows opposite to their predecessor (csrc/traversal.h: start in
    what the memory-side cache still holds).  Their results are functions of the row / molecule alone: four launches in a
    row (ascending, descending, ...) are bit-identical, ragged last tiles included, and equal the fp64 contraction."""
    from druggen_amd import functional as dgf
    x = _gen((R, 128), 300).float().cuda()
    res = _gen((R, 128), 301).float().cuda()
    w1, w2, w3 = (_gen((384, 128), 302) * 0.1).float().cuda(), (_gen((128, 384), 303) * 0.1).float().cuda(), (_gen((128, 128), 304) * 0.1).float().cuda()
    b1, g, be = _gen((384,), 305).float().cuda(), (_gen((128,), 306) * 0.1 + 1).float().cuda(), _gen((128,), 307).float().cuda()
    runs = []
    for _ in range(4):
        h, bits = dgf.row_gemm(x, dgf.packed_weight(w1, 0), 128, 384, bias=b1, relu=True, want_relu_bits=True)
        y, mean, rstd, pre = dgf.row_gemm(h, dgf.packed_weight(w2, 0), 384, 128, residual=res, ln=(g, be, 1e-5), want_pre=True)
        z = dgf.row_gemm(y, dgf.packed_weight(w3, 0), 128, 128, residual=x)
        dh = dgf.row_gemm(z, dgf.packed_weight(w2, 1), 128, 384, mask_bits=bits)
        runs.append([h, y, mean, rstd, pre, z, dh])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)
    hd = torch.relu(x.double().cpu() @ w1.double().cpu().t() + b1.double().cpu())
    assert _rel(runs[0][0], hd) < TOL
    assert _rel(runs[0][4], hd @ w2.double().cpu().t() + res.double().cpu()) < TOL
    B, N, C = (R + 44) // 45 // 45, 45, 128      # attention core: molecules in ascending / descending order
    f = lambda shape, s_: _gen(shape, s_).float().cuda()
    q, k, v, e = f((B, N, C), 310), f((B, N, C), 311), f((B, N, C), 312), f((B, N, N, C), 313)
    outs = [dgf.attn_core(q, k, v, e, 0.25) for _ in range(4)]
    for s_, o in outs[1:]:
        assert torch.equal(s_, outs[0][0]) and torch.equal(o, outs[0][1])


@pytest.mark.parametrize("B,N", [(1, 1), (2, 9), (3, 45), (5, 48), (37, 45), (2, 49), (3, 90), (2, 96), (9, 64), (260, 50), (1, 96)])
def test_fused_float32_attention_half_forward(B, N):
    """dg_attn_half_f32_fwd -- e = y We^T + be, sc = alpha q_i k_j (e + 1) e, o_i = sum_j softmax_j(sc) v_j,
    y2 = LN(y + sc Woe^T + boe) (reference layers.py:114-135, 186-188) as ONE launch -- against the fp64 closed form and
    against the three launches it replaces; with and without the outputs only a backward reads; repeated launches
    (ascending / descending traversal) bit-identical.  N > 48 (BASELINE configs[4]: N = 90): two stages per row group with an
    online softmax across them."""
    from druggen_amd import functional as dgf
    lib = _lib().load()
    C, alpha, eps = 128, 0.25, 1e-5
    y = _gen((B, N, N, C), 400).float().cuda()
    q, k, v = (_gen((B, N, C), 401 + i).float().cuda() for i in range(3))
    we, woe = (_gen((C, C), 404) * 0.1).float().cuda(), (_gen((C, C), 405) * 0.1).float().cuda()
    be, boe = (_gen((C,), 406) * 0.1).float().cuda(), (_gen((C,), 407) * 0.1).float().cuda()
    g4, b4 = (_gen((C,), 408) * 0.1 + 1).float().cuda(), (_gen((C,), 409) * 0.1).float().cuda()
    R = B * N * N
    pe, po = dgf.packed_weight(we, 0), dgf.packed_weight(woe, 0)
    st = torch.cuda.current_stream().cuda_stream

    def run(keep):
        e, s, pre = ((torch.full((R, C), float("nan"), device="cuda") for _ in range(3)) if keep else (None, None, None))
        y2, o = torch.empty(R, C, device="cuda"), torch.empty(B, N, C, device="cuda")
        mean, rstd = torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
        p = lambda t: None if t is None else t.data_ptr()
        _lib().check(lib.dg_attn_half_f32_fwd(y.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), pe.data_ptr(), be.data_ptr(),
                                              po.data_ptr(), boe.data_ptr(), g4.data_ptr(), b4.data_ptr(), p(e), p(s), o.data_ptr(),
                                              y2.data_ptr(), p(pre), mean.data_ptr(), rstd.data_ptr(), B, N, C, alpha, eps, st),
                     "dg_attn_half_f32_fwd")
        return e, s, o, y2, pre, mean, rstd

    e, s, o, y2, pre, mean, rstd = run(True)
    yd = y.double().cpu().reshape(R, C)
    ed = yd @ we.double().cpu().t() + be.double().cpu()
    e4 = ed.view(B, N, N, C)
    sc = alpha * q.double().cpu()[:, :, None, :] * k.double().cpu()[:, None, :, :] * (e4 * e4 + e4)
    od = (torch.softmax(sc, dim=2) * v.double().cpu()[:, None, :, :]).sum(2)
    pred = yd + sc.reshape(R, C) @ woe.double().cpu().t() + boe.double().cpu()
    mu, var = pred.mean(1, keepdim=True), pred.var(1, unbiased=False, keepdim=True)
    y2d = (pred - mu) / torch.sqrt(var + eps) * g4.double().cpu() + b4.double().cpu()
    for got, want in ((e, ed), (s, sc.reshape(R, C)), (o, od), (pre, pred), (y2, y2d), (mean, mu.squeeze(1)),
                      (rstd, 1 / torch.sqrt(var.squeeze(1) + eps))):
        assert _rel(got, want) < TOL
    # the three launches it replaces
    e3 = dgf.row_gemm(y.view(R, C), pe, C, C, bias=be)
    s3, o3 = torch.empty_like(e3), torch.empty(B, N, C, device="cuda")
    _lib().check(lib.dg_attn_core_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), e3.data_ptr(), s3.data_ptr(), o3.data_ptr(), B, N, C,
                                      alpha, 0, st), "dg_attn_core_fwd")
    y3, _, _, p3 = dgf.row_gemm(s3, po, C, C, bias=boe, residual=y.view(R, C), ln=(g4, b4, eps), want_pre=True)
    for got, want in ((e, e3), (s, s3), (o, o3), (pre, p3), (y2, y3)):
        assert _rel(got, want.double().cpu()) < 5e-6
    # outputs only a backward reads may be skipped; launches are bit-reproducible in either traversal direction
    for keep in (False, True, True):
        again = run(keep)
        for a_, b_ in zip((e, s, o, y2, pre, mean, rstd), again):
            if b_ is not None:
                assert torch.equal(a_, b_)
    assert lib.dg_attn_half_f32_fwd(y.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), pe.data_ptr(), be.data_ptr(),
                                    po.data_ptr(), boe.data_ptr(), g4.data_ptr(), b4.data_ptr(), None, None, o.data_ptr(),
                                    y2.data_ptr(), None, mean.data_ptr(), rstd.data_ptr(), B, 97, C, alpha, eps, st) != 0


@pytest.mark.parametrize("B,N", [(1, 1), (2, 9), (3, 45), (5, 48), (37, 45), (300, 9)])
def test_fused_float32_attention_half_backward_part1(B, N):
    """dg_attn_half_f32_bwd1 -- ln4 backward, ds = dz4 Woe, attention-core backward (reference layers.py:119-135, 186-188,
    differentiated) as ONE launch -- against fp64 autograd of the same expressions and against the two launches it
    replaces; more molecules than workgroups (300 > 256); repeated launches bit-identical."""
    from druggen_amd import functional as dgf
    lib = _lib().load()
    C, alpha, eps = 128, 0.25, 1e-5
    R = B * N * N
    dy2 = _gen((R, C), 500).float().cuda()
    pre = (_gen((R, C), 501) * 2 + 0.3).float().cuda()
    e = (_gen((B, N, N, C), 502) * 0.5).float().cuda()
    q, k, v, d_o = (_gen((B, N, C), 503 + i).float().cuda() for i in range(4))
    woe = (_gen((C, C), 507) * 0.1).float().cuda()
    g4 = (_gen((C,), 508) * 0.1 + 1).float().cuda()
    mean = pre.double().mean(1).float().contiguous()
    rstd = (1 / torch.sqrt(pre.double().var(1, unbiased=False) + eps)).float().contiguous()
    pwo = dgf.packed_weight(woe, 1)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(int(lib.dg_attn_half_f32_bwd1_workspace_bytes(B)), dtype=torch.uint8, device="cuda")

    def run(ds=None):
        dz, de = (torch.full((R, C), float("nan"), device="cuda") for _ in range(2))
        dq, dk, dv = (torch.full((B, N, C), float("nan"), device="cuda") for _ in range(3))
        dgb = torch.full((2, C), float("nan"), device="cuda")
        _lib().check(lib.dg_attn_half_f32_bwd1(dy2.data_ptr(), pre.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g4.data_ptr(),
                                               pwo.data_ptr(), e.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                               d_o.data_ptr(), dz.data_ptr(), None if ds is None else ds.data_ptr(),
                                               de.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                               dv.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), ws.data_ptr(), ws.numel(),
                                               B, N, C, alpha, st), "dg_attn_half_f32_bwd1")
        return dz, de, dq, dk, dv, dgb[0], dgb[1]

    got = run()
    # fp64 autograd: pre = p0 + sc Woe^T with p0 chosen so that pre is the given pre-LayerNorm sum
    leaf = lambda t: t.double().cpu().requires_grad_(True)
    ed, qd, kd, vd, gd = leaf(e), leaf(q), leaf(k), leaf(v), leaf(g4)
    bd = torch.zeros(C, dtype=torch.float64, requires_grad=True)
    sc = alpha * qd[:, :, None, :] * kd[:, None, :, :] * (ed * ed + ed)
    od = (torch.softmax(sc, dim=2) * vd[:, None, :, :]).sum(2)
    wd = woe.double().cpu()
    p0 = (pre.double().cpu() - sc.detach().reshape(R, C) @ wd.t()).requires_grad_(True)
    pd = p0 + sc.reshape(R, C) @ wd.t()
    y2 = torch.nn.functional.layer_norm(pd, (C,), gd, bd, eps)
    loss = (y2 * dy2.double().cpu()).sum() + (od * d_o.double().cpu()).sum()
    want = torch.autograd.grad(loss, [p0, ed, qd, kd, vd, gd, bd])
    for a_, b_ in zip(got, want):
        assert _rel(a_.reshape(b_.shape), b_) < TOL
    # the two launches it replaces
    dz_r, ds_r, dg_r, db_r = dgf.ln_bwd_row_gemm(pre, g4, mean, rstd, dy2, pwo, want_affine=True)
    dq_r, dk_r, dv_r, de_r = dgf._attn_bwd_launch(q, k, v, e, ds_r.view(B, N, N, C), d_o, alpha)
    for a_, b_ in zip(got, (dz_r, de_r, dq_r, dk_r, dv_r, dg_r, db_r)):
        assert _rel(a_.reshape(b_.shape), b_.double().cpu()) < 5e-6
    for _ in range(2):
        for a_, b_ in zip(got, run()):
            assert torch.equal(a_, b_)
    # the instance that also writes ds (a pass whose second order will read it): same results, ds = dz4 Woe
    ds = torch.full((R, C), float("nan"), device="cuda")
    for a_, b_ in zip(got, run(ds)):
        assert torch.equal(a_, b_)
    assert _rel(ds, want[0] @ wd) < TOL and _rel(ds, ds_r.double().cpu()) < 5e-6
    assert lib.dg_attn_half_f32_bwd1(dy2.data_ptr(), pre.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g4.data_ptr(),
                                     pwo.data_ptr(), e.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), d_o.data_ptr(),
                                     got[0].data_ptr(), None, got[1].data_ptr(), got[2].data_ptr(), got[3].data_ptr(),
                                     got[4].data_ptr(), got[5].data_ptr(), got[6].data_ptr(), ws.data_ptr(), 16, B, N, C,
                                     alpha, st) != 0      # workspace too small


def test_attn_core_is_bit_reproducible():
    from druggen_amd import functional as dgf
    B, N, C, alpha = 4, 45, 128, 0.25
    f = lambda shape, s: _gen(shape, s).float().cuda().requires_grad_(True)
    q, k, v, e = f((B, N, C), 1), f((B, N, C), 2), f((B, N, C), 3), f((B, N, N, C), 4)
    ws, wo = _gen((B, N, N, C), 5).float().cuda(), _gen((B, N, C), 6).float().cuda()
    outs = []
    for _ in range(3):
        s, o = dgf.attn_core(q, k, v, e, alpha)
        g = torch.autograd.grad([s, o], [q, k, v, e], [ws, wo])
        outs.append([s.detach().clone(), o.detach().clone()] + [x.clone() for x in g])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_attn_core_rejects_unsupported_shapes():
    lib = _lib().load()
    x = torch.zeros(4, device="cuda")
    st = lib.dg_attn_core_fwd(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                              1, 200, 128, 0.25, 0, None)
    assert st == -1 and b"unsupported shape" in lib.dg_last_error_string()
    st = lib.dg_attn_core_fwd(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                              1, 9, 6, 0.25, 0, None)
    assert st == -1
    st = lib.dg_attn_core_fwd(None, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                              1, 9, 8, 0.25, 0, None)
    assert st == -2


def test_attn_core_full_size_properties():
    """BASELINE configs[1] shape (B=256, N=45, C=128): properties that need no
    oracle run.  v == 1 -> o == 1 (softmax rows sum to one); o is linear in v;
    s does not depend on v."""
    from druggen_amd import functional as dgf
    B, N, C, alpha = 256, 45, 128, 0.25
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k = (torch.randn(B, N, C, device="cuda", generator=g) for _ in range(2))
    v1, v2 = (torch.randn(B, N, C, device="cuda", generator=g) for _ in range(2))
    e = torch.randn(B, N, N, C, device="cuda", generator=g) * 0.5
    s1, o1 = dgf.attn_core(q, k, v1, e, alpha)
    s2, o2 = dgf.attn_core(q, k, v2, e, alpha)
    _, o12 = dgf.attn_core(q, k, v1 + 2 * v2, e, alpha)
    _, ones = dgf.attn_core(q, k, torch.ones_like(v1), e, alpha)
    assert torch.equal(s1, s2)
    assert (ones - 1).abs().max().item() < 1e-5
    assert (o12 - (o1 + 2 * o2)).abs().max().item() < 1e-4 * max(1.0, o12.abs().max().item())
    # s against the definition (elementwise, cheap on the GPU itself)
    s_def = alpha * q.unsqueeze(2) * k.unsqueeze(1) * (e * e + e)
    assert (s1 - s_def).abs().max().item() <= 1e-5 * s_def.abs().max().item()


LN_SHAPES = [(7, 8), (33, 16), (5, 12), (1000, 128), (3, 384), (257, 32), (65, 1024), (2, 64)]


@pytest.mark.parametrize("R,C", LN_SHAPES)
@pytest.mark.parametrize("with_residual", [True, False])
def test_ln_residual_all_orders(R, C, with_residual):
    from druggen_amd import functional as dgf
    a, r = _gen((R, C), 1), _gen((R, C), 2)
    gamma, beta = 1 + 0.2 * _gen((C,), 3), _gen((C,), 4)
    dy, tz = _gen((R, C), 5), _gen((R, C), 6)
    z = a + r if with_residual else a
    y_ref, mu, rstd = km.ln_fwd(z, gamma, beta)
    dz_ref, dg_ref, db_ref = km.ln_bwd(z, gamma, mu, rstd, dy)
    gz_ref, gg_ref, gdy_ref = km.ln_bwd2(z, gamma, mu, rstd, dy, tz)
    f = lambda x: x.float().cuda().requires_grad_(True)
    ad, rd, gd, bd, dyd = map(f, (a, r, gamma, beta, dy))
    y = dgf.ln_residual(ad, rd if with_residual else None, gd, bd)
    assert _rel(y, y_ref) < TOL
    ins = [ad, gd, bd] + ([rd] if with_residual else [])
    grads = torch.autograd.grad(y, ins, dyd, create_graph=True)
    assert _rel(grads[0], dz_ref) < TOL
    assert _rel(grads[1], dg_ref) < TOL and _rel(grads[2], db_ref) < TOL
    if with_residual:
        assert torch.equal(grads[3], grads[0])
    phi = (grads[0] * tz.float().cuda()).sum()
    gz, gg, gdy = torch.autograd.grad(phi, [ad, gd, dyd])
    assert _rel(gz, gz_ref) < 5 * TOL and _rel(gg, gg_ref) < 5 * TOL and _rel(gdy, gdy_ref) < 5 * TOL


def test_ln_matches_torch_layer_norm_at_edge_tensor_size():
    """configs[1] edge tensor rows (B*N*N = 518400, C=128) vs torch on the same GPU."""
    from druggen_amd import functional as dgf
    R, C = 256 * 45 * 45, 128
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(R, C, device="cuda", generator=g)
    r = torch.randn(R, C, device="cuda", generator=g)
    gamma = 1 + 0.1 * torch.randn(C, device="cuda", generator=g)
    beta = torch.randn(C, device="cuda", generator=g)
    y = dgf.ln_residual(a, r, gamma, beta)
    want = torch.nn.functional.layer_norm(a + r, (C,), gamma, beta, 1e-5)
    assert (y - want).abs().max().item() < 2e-5


WGRAD_SHAPES = [(2025 * 2, 64, 5), (300, 64, 13), (1000, 5, 128), (2025 * 3, 13, 128), (70, 10, 128), (33, 1, 64), (513, 16, 32), (5, 32, 32), (50, 32, 64), (77, 128, 64), (1000, 128, 128), (4097, 384, 128), (333, 128, 384),
                (2025 * 3, 128, 128), (64, 64, 64), (129, 96, 32), (31, 32, 96), (40, 64, 32)]


@pytest.mark.parametrize("R,N,K", WGRAD_SHAPES)
@pytest.mark.parametrize("bias", [True, False])
def test_linear_wgrad_kernel(R, N, K, bias):
    from druggen_amd import functional as dgf
    dy, x = _gen((R, N), 1), _gen((R, K), 2)
    dw_ref, db_ref = dy.t() @ x, dy.sum(0)
    lib = _lib().load()
    assert lib.dg_linear_wgrad_workspace_bytes(R, N, K) > 0 or lib.dg_linear_wgrad_workspace_bytes(R, K, N) > 0
    dw, db = dgf._wgrad(dy.float().cuda(), x.float().cuda(), bias)
    assert _rel(dw, dw_ref) < TOL
    if bias:
        assert _rel(db, db_ref) < TOL
    else:
        assert db is None
    dw2, _ = dgf._wgrad(dy.float().cuda(), x.float().cuda(), bias)
    assert torch.equal(dw, dw2)          # fixed-order split-K reduction


def test_linear_wgrad_full_size_matches_library_gemm():
    """configs[1] edge rows (R = 518400): fc1 weight gradient vs the BLAS path."""
    from druggen_amd import functional as dgf
    R, N, K = 256 * 45 * 45, 384, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    dy = torch.randn(R, N, device="cuda", generator=g)
    x = torch.randn(R, K, device="cuda", generator=g)
    dw, db = dgf._wgrad(dy, x, True)
    ref = dy.double().t() @ x.double()
    assert ((dw.double() - ref).norm() / ref.norm()).item() < 1e-5
    assert ((db.double() - dy.double().sum(0)).norm() / dy.double().sum(0).norm()).item() < 1e-5


@pytest.mark.parametrize("shape,N,K", [((3, 7, 7, 32), 64, 32), ((2, 9, 128), 128, 128), ((4, 5, 5, 16), 24, 16)])
def test_linear_function_first_and_second_order(shape, N, K):
    """dgf.linear == F.linear through double backward (supported and fallback shapes)."""
    from druggen_amd import functional as dgf
    x = _gen(shape, 1).float().cuda().requires_grad_(True)
    w = (_gen((N, K), 2) * 0.2).float().cuda().requires_grad_(True)
    b = _gen((N,), 3).float().cuda().requires_grad_(True)
    dy = _gen(shape[:-1] + (N,), 4).float().cuda().requires_grad_(True)
    tx = _gen(shape, 5).float().cuda()
    outs = []
    for fn in (dgf.linear, torch.nn.functional.linear):
        y = fn(x, w, b)
        gx, gw, gb = torch.autograd.grad(y, [x, w, b], dy, create_graph=True)
        second = torch.autograd.grad((gx * tx).sum(), [w, dy])
        outs.append([y, gx, gw, gb, *second])
    for a, r in zip(*outs):
        assert _rel(a, r.double().cpu()) < TOL


@pytest.mark.parametrize("R", [1, 63, 64, 200, 4097])
@pytest.mark.parametrize("K,N", [(128, 128), (128, 384), (384, 128)])
def test_row_gemm_forward_and_dgrad_modes(R, K, N):
    from druggen_amd import functional as dgf
    a = _gen((R, K), 1)
    w = _gen((N, K), 2) * 0.1          # nn.Linear layout [out, in]
    b = _gen((N,), 3)
    ad, wd, bd = a.float().cuda(), w.float().cuda(), b.float().cuda()
    y = dgf.row_gemm(ad, dgf.packed_weight(wd, 0), K, N, bias=bd)
    assert _rel(y, a @ w.t() + b) < TOL
    y = dgf.row_gemm(ad, dgf.packed_weight(wd, 0), K, N, bias=bd, relu=True)
    assert _rel(y, torch.relu(a @ w.t() + b)) < TOL
    # input gradient: dx[R,K'] = dy[R,N'] @ W[N',K']  (contraction over W's rows)
    w2 = _gen((K, N), 4) * 0.1         # a Linear(N -> K) weight, used transposed
    y = dgf.row_gemm(ad, dgf.packed_weight(w2.float().cuda(), 1), K, N)
    assert _rel(y, a @ w2) < TOL


@pytest.mark.parametrize("need_edge", [True, False])
def test_encoder_runs_ln6_backward_inside_the_next_blocks_dy_gemm(need_edge):
    """TransformerEncoder in float32: block l + 1 receives the handle of block l's ln6 and, in a plain first-order
    backward, runs that LayerNorm's backward as the epilogue of its own dy GEMM (dg_row_gemm_ln_bwd): one edge-level
    LayerNorm-backward launch less per block boundary, same gradients as with the separate launches
    (options.ln_bwd_epilogue = False) to fp32 round-off.  Under create_graph (the gradient penalty's first pass,
    loss.py:32-39) the separate, twice-differentiable launches stay."""
    import os
    from druggen_amd import functional as dgf
    from druggen_amd.model.layers import TransformerEncoder
    L = _lib()
    torch.manual_seed(11)
    B, N, C, depth = 2, 9, 128, 3
    enc = TransformerEncoder(C, depth, 8, torch.nn.ReLU, mlp_ratio=3, drop_rate=0.0).cuda()
    x = torch.randn(B, N, C, device="cuda", requires_grad=True)
    y = (0.5 * torch.randn(B, N, N, C, device="cuda")).requires_grad_(True)
    gx = torch.randn(B, N, C, device="cuda")
    gy = torch.randn(B, N, N, C, device="cuda")
    params = [p for p in enc.parameters()]

    def grads():
        xo, yo = enc(x, y, need_edge)
        outs, gos = ([xo, yo], [gx, gy]) if need_edge else ([xo], [gx])
        return torch.autograd.grad(outs, [x, y] + params, gos, allow_unused=True)

    def count(fn):
        L.prof_enable(True, kernels=["ln_bwd"])
        L.prof_reset()
        out = fn()
        n = L.prof_read("ln_bwd")[0]
        L.prof_enable(False)
        return out, n

    fused, n_fused = count(grads)
    from druggen_amd.options import options
    with options.override(ln_bwd_epilogue=False):
        plain, n_plain = count(grads)
    assert n_plain - n_fused == depth - 1, (n_plain, n_fused)      # one ln6 backward per block boundary
    for a_, b_ in zip(fused, plain):
        assert (a_ is None) == (b_ is None)
        if a_ is not None:
            # (same forward, same masks; the two backward paths round dz differently in the last bit, and a handful of the 18 x 384
            # elements of the node feed-forward's dh then land on the other side of an fp16 rounding boundary: 2.6e-5 measured)
            assert _rel(a_, b_.double().cpu()) < 5e-5
    # twice-differentiable pass: no epilogue fusion, and the result can be differentiated again
    xo, yo = enc(x, y, need_edge)
    with dgf.inputs_only_backward():
        (g1,), n_cg = count(lambda: torch.autograd.grad([xo], [y], [gx], create_graph=True))
    assert n_cg > 0 and g1.requires_grad


@pytest.mark.parametrize("with_residual", [True, False])
@pytest.mark.parametrize("R", [1, 63, 64, 200, 4097, 20000])
@pytest.mark.parametrize("K", [128])
def test_row_gemm_with_layernorm_backward_epilogue(R, K, with_residual):
    """dg_row_gemm_ln_bwd: the input-gradient GEMM that produces the gradient of a LayerNorm output, with that
    LayerNorm's backward (layers.py:187-192) as its epilogue, against float64 autograd of
    LayerNorm(pre) . (a W + residual) and against the two launches it replaces (row GEMM, then dg_ln_residual_bwd)."""
    from druggen_amd import functional as dgf
    a = _gen((R, K), 1)
    w = _gen((K, 128), 2) * 0.1          # a Linear(128 -> K) weight, used transposed: dx = dy @ w
    res = _gen((R, 128), 3) if with_residual else None
    pre = (_gen((R, 128), 4) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * _gen((128,), 5)).requires_grad_(True)
    beta = _gen((128,), 6).requires_grad_(True)
    v = a @ w + (res if with_residual else 0)
    out = torch.nn.functional.layer_norm(pre, (128,), gamma, beta, 1e-5)
    dz_ref, dg_ref, db_ref = torch.autograd.grad(out, [pre, gamma, beta], v)
    pre_d = pre.detach().float().cuda()
    mean = pre_d.mean(-1)
    rstd = (pre_d.var(-1, unbiased=False) + 1e-5).rsqrt()
    packed = dgf.packed_weight(w.float().cuda(), 1)
    ad = a.float().cuda()
    resd = res.float().cuda() if with_residual else None
    gd = gamma.detach().float().cuda()
    dz, dg, db = dgf.row_gemm_ln_bwd(ad, packed, K, resd, pre_d, gd, mean, rstd)
    assert _rel(dz, dz_ref) < TOL and _rel(dg, dg_ref) < TOL and _rel(db, db_ref) < TOL
    # the two launches it replaces
    dy = dgf.row_gemm(ad, packed, K, 128, residual=resd)
    dz2, dg2, db2 = dgf._ln_bwd_rows(pre_d, gd, mean, rstd, dy)
    assert _rel(dz, dz2.double().cpu()) < 1e-5 and _rel(dg, dg2.double().cpu()) < 1e-5 and _rel(db, db2.double().cpu()) < 1e-5
    again = dgf.row_gemm_ln_bwd(ad, packed, K, resd, pre_d, gd, mean, rstd)
    assert all(torch.equal(x, y) for x, y in zip((dz, dg, db), again))      # fixed-order partial sums


@pytest.mark.parametrize("R", [1, 63, 64, 65, 200, 4097, 16384 + 17, 70000])
def test_row_gemm_with_layernorm_backward_prologue(R):
    """dg_row_gemm_ln_bwd_in: the input-gradient GEMM whose A operand is the INPUT gradient of a LayerNorm
    (layers.py:187-190 backward: ln4 feeds the out_e input gradient), with that LayerNorm's backward run by the GEMM's
    producer waves -- against float64 autograd of LayerNorm(pre) and dz @ W, and against the two launches it replaces
    (dg_ln_residual_bwd, then the row GEMM).  Ragged row counts: the last tile clamps its addresses."""
    from druggen_amd import functional as dgf
    dy = _gen((R, 128), 1)
    w = _gen((128, 128), 2) * 0.1          # a Linear(128 -> 128) weight, used transposed: ds = dz @ w
    pre = (_gen((R, 128), 4) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * _gen((128,), 5)).requires_grad_(True)
    beta = _gen((128,), 6).requires_grad_(True)
    out = torch.nn.functional.layer_norm(pre, (128,), gamma, beta, 1e-5)
    dz_ref, dg_ref, db_ref = torch.autograd.grad(out, [pre, gamma, beta], dy)
    y_ref = dz_ref @ w
    pre_d = pre.detach().float().cuda()
    mean = pre_d.mean(-1)
    rstd = (pre_d.var(-1, unbiased=False) + 1e-5).rsqrt()
    packed = dgf.packed_weight(w.float().cuda(), 1)
    dyd = dy.float().cuda()
    gd = gamma.detach().float().cuda()
    dz, y, dg, db = dgf.ln_bwd_row_gemm(pre_d, gd, mean, rstd, dyd, packed)
    assert _rel(dz, dz_ref) < TOL and _rel(y, y_ref) < TOL and _rel(dg, dg_ref) < TOL and _rel(db, db_ref) < TOL
    dz2, dg2, db2 = dgf._ln_bwd_rows(pre_d, gd, mean, rstd, dyd)
    y2 = dgf.row_gemm(dz2, packed, 128, 128)
    for got, want in ((dz, dz2), (y, y2), (dg, dg2), (db, db2)):
        assert _rel(got, want.double().cpu()) < 1e-5
    again = dgf.ln_bwd_row_gemm(pre_d, gd, mean, rstd, dyd, packed)
    assert all(torch.equal(a_, b_) for a_, b_ in zip((dz, y, dg, db), again))      # fixed-order partial sums
    # input-gradient-only passes: no dgamma / dbeta
    dz3, y3, none_g, none_b = dgf.ln_bwd_row_gemm(pre_d, gd, mean, rstd, dyd, packed, want_affine=False)
    assert none_g is None and none_b is None and torch.equal(dz3, dz) and torch.equal(y3, y)


def test_attn_block_backward_with_and_without_the_layernorm_prologue():
    """The edge-level ln4 backward inside the out_e input-gradient GEMM (options.ln_bwd_prologue, default on) gives the
    gradients of the separate launches: one attention block at an edge-level row count, all parameter and input grads."""
    import os
    from druggen_amd.model.layers import Encoder_Block
    torch.manual_seed(3)
    B, N, C = 40, 45, 128                      # 40 * 45 * 45 = 81 000 rows >= DG_EDGE_ROWS
    blk = Encoder_Block(C, 8, torch.nn.ReLU(), 3, 0.0).cuda()
    x = torch.randn(B, N, C, device="cuda").requires_grad_(True)
    y = torch.randn(B, N, N, C, device="cuda").requires_grad_(True)
    px, py = torch.randn(B, N, C, device="cuda"), torch.randn(B, N, N, C, device="cuda")

    from druggen_amd.options import options

    def run(flag):
        with options.override(ln_bwd_prologue=(flag == "on")):
            for p in blk.parameters():
                p.grad = None
            xo, yo = blk(x, y)
            gx, gy = torch.autograd.grad((xo * px).sum() + (yo * py).sum(), [x, y], retain_graph=True)
            ((xo * px).sum() + (yo * py).sum()).backward()
        return [gx, gy] + [p.grad.clone() for p in blk.parameters() if p.grad is not None]

    on, off = run("on"), run("off")
    assert len(on) == len(off)
    for a_, b_ in zip(on, off):
        assert _rel(a_, b_.double().cpu()) < 2e-5


@pytest.mark.parametrize("N,K", [(128, 128), (384, 128), (128, 384)])
def test_wgrad_split_bf16_is_fp32_class_accurate(N, K):
    """Same claim for the weight-gradient kernel (dW = dy^T x on the fp16 hi + lo split with running column scales):
    its error against fp64, relative to sum_r |dy x|, stays at the level of torch's fp32 matmul of the same data."""
    from druggen_amd import functional as dgf
    R = 8192
    dy, x = _gen((R, N), 21), _gen((R, K), 22)
    dyd, xd = dy.float().cuda(), x.float().cuda()
    want = dy.t() @ x
    scale = dy.abs().t() @ x.abs()
    def err(w):
        e = (w.double().cpu() - want).abs() / scale
        return e.max().item(), e.pow(2).mean().sqrt().item()
    mine_max, mine_rms = err(dgf._wgrad(dyd, xd, True)[0])
    torch.backends.cuda.matmul.allow_tf32 = False
    ref_max, ref_rms = err(dyd.t() @ xd)
    print(f"N={N} K={K}: wgrad max {mine_max:.2e} rms {mine_rms:.2e} | fp32 matmul max {ref_max:.2e} rms {ref_rms:.2e}")
    assert mine_rms < 1.5 * ref_rms + 2e-9 and mine_max < 2.5 * ref_max


WGRAD_RANGE_CASES = ["tiny_gradients", "huge_activations", "growing_rows", "shrinking_rows", "column_scales", "heavy_tailed",
                     "zero_columns", "denormals", "one_spike"]


@pytest.mark.parametrize("case", WGRAD_RANGE_CASES)
@pytest.mark.parametrize("N,K", [(128, 128), (384, 128), (128, 384)])
def test_wgrad_running_column_scales_hold_fp32_accuracy_over_the_fp32_range(N, K, case):
    """The fp16 operands of the weight-gradient kernel carry one running power-of-two scale per COLUMN of dy and of x
    (csrc/linear_wgrad.hip, SPLIT 2).  Data that forces the scales to move -- gradients of 1e-20, activations of 1e15,
    rows that grow or shrink by 2^40 through the launch, columns 2^30 apart, Cauchy tails, all-zero columns, fp32
    denormals, one spike 2^25 above the rest of its column -- must give the accuracy of an fp32 matmul of the same data:
    error against fp64, relative to sum_r |dy x| per output, no worse than 2 x torch's fp32 result (+ 2^-24)."""
    from druggen_amd import functional as dgf
    R = 6000
    g = torch.Generator().manual_seed(N + K + len(case))
    dy = torch.randn(R, N, generator=g, dtype=torch.float64)
    x = torch.randn(R, K, generator=g, dtype=torch.float64)
    ramp = torch.linspace(-20, 20, R, dtype=torch.float64)[:, None]
    if case == "tiny_gradients":
        dy = dy * 1e-20
    elif case == "huge_activations":
        x = x * 1e15
    elif case == "growing_rows":
        dy, x = dy * torch.exp2(ramp), x * torch.exp2(0.5 * ramp)
    elif case == "shrinking_rows":
        dy, x = dy * torch.exp2(-ramp), x * torch.exp2(-0.5 * ramp)
    elif case == "column_scales":
        dy = dy * torch.exp2(torch.randint(-15, 16, (1, N), generator=g).double())
        x = x * torch.exp2(torch.randint(-15, 16, (1, K), generator=g).double())
    elif case == "heavy_tailed":
        dy = dy / torch.randn(R, N, generator=g, dtype=torch.float64).abs().clamp_min(1e-4)
        x = x / torch.randn(R, K, generator=g, dtype=torch.float64).abs().clamp_min(1e-4)
    elif case == "zero_columns":
        dy[:, ::3] = 0
        x[:, 1::4] = 0
        dy[: R // 2, 1] = 0          # a column that starts only half way through the rows
    elif case == "denormals":
        dy = dy * 1e-39
    elif case == "one_spike":
        dy[R // 3, :] *= 2.0 ** 25
        x[2 * R // 3, :] *= 2.0 ** 25
    dyd, xd = dy.float().cuda(), x.float().cuda()
    dy, x = dyd.double().cpu(), xd.double().cpu()          # the fp32-rounded operands are the ground truth's inputs
    want = dy.t() @ x
    scale = (dy.abs().t() @ x.abs()).clamp_min(1e-300)
    dw, db = dgf._wgrad(dyd, xd, True)
    assert torch.isfinite(dw).all()
    torch.backends.cuda.matmul.allow_tf32 = False
    ref = (dyd.t() @ xd).double().cpu()
    e_mine = ((dw.double().cpu() - want).abs() / scale)
    e_ref = ((ref - want).abs() / scale)
    assert e_mine.max().item() <= 2 * e_ref.max().item() + 2.0 ** -24, (case, e_mine.max().item(), e_ref.max().item())
    assert e_mine.pow(2).mean().sqrt().item() <= 2 * e_ref.pow(2).mean().sqrt().item() + 1e-8
    zero = scale < 1e-290
    assert (dw.double().cpu()[zero] == 0).all()          # all-zero columns give exact zeros
    assert _rel(db, dy.sum(0)) < 1e-5 or case in ("heavy_tailed", "one_spike")


def test_wgrad_propagates_inf_and_nan_like_a_matmul():
    """dy^T x with an inf / a NaN in one row: the reference's `mm` puts non-finite values in exactly the outputs that row
    feeds; the split kernel must not spread them to other outputs or swallow them.  As in the row GEMM (below), an inf
    operand comes out as NaN (hi = inf, lo = inf - inf) where `mm` gives +-inf: non-finite either way."""
    from druggen_amd import functional as dgf
    R, N, K = 700, 128, 128
    g = torch.Generator().manual_seed(9)
    dy, x = torch.randn(R, N, generator=g), torch.randn(R, K, generator=g)
    dy[100, 5] = float("inf")
    dy[400, 77] = float("nan")
    x[600, 9] = float("-inf")
    dyd, xd = dy.cuda(), x.cuda()
    dw, _ = dgf._wgrad(dyd, xd, True)
    want = dy.double().t() @ x.double()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(dw).cpu(), fin)
    assert torch.isnan(dw).cpu()[torch.isnan(want)].all()
    assert _rel(dw.cpu()[fin], want[fin]) < TOL


@pytest.mark.parametrize("data", ["normal", "scaled_rows_and_columns", "heavy_tailed"])
@pytest.mark.parametrize("K,N", [(128, 128), (128, 384), (384, 128)])
def test_row_gemm_split_bf16_is_fp32_class_accurate(K, N, data):
    """The row GEMM splits its fp32 operands into 16-bit planes and runs MFMA cross products with fp32
    accumulation: every 128-wide row chunk and every weight column is scaled by a power of two, split into two fp16
    planes, three products per k-step (K = 384: one scale per (row, 128-chunk), each chunk accumulated on its own and
    folded with its inverse scale).  It is not a reduced-precision GEMM: measured against fp64,
    the element-wise error (relative to sum_k |a_k w_k|, the natural scale of a dot product) must be at the level
    of an fp32 GEMM of the same data -- torch's fp32 matmul here -- and orders of magnitude below bf16.  The
    scaled cases multiply rows by 2^-60..2^60 and weight columns by 1e-6..1e6 (the per-row / per-column scales
    must absorb them exactly); the heavy-tailed case puts a 1e4 dynamic range inside every row."""
    from druggen_amd import functional as dgf
    R = 4096
    a = _gen((R, K), 11)
    w = _gen((N, K), 12) * 0.1
    if data == "scaled_rows_and_columns":
        g = torch.Generator().manual_seed(5)
        a = a * torch.exp2(torch.randint(-60, 61, (R, 1), generator=g).double())
        w = w * (10.0 ** (torch.rand(N, 1, generator=g, dtype=torch.float64) * 12 - 6))
    elif data == "heavy_tailed":
        g = torch.Generator().manual_seed(6)
        a = a * torch.exp(torch.randn(R, K, generator=g, dtype=torch.float64) * 2.5)
        w = w * torch.exp(torch.randn(N, K, generator=g, dtype=torch.float64) * 2.5)
    ad, wd = a.float().cuda(), w.float().cuda()
    a, w = ad.double().cpu(), wd.double().cpu()          # the fp32 operands, exactly
    want = a @ w.t()
    scale = a.abs() @ w.abs().t()
    def err(y):
        e = (y.double().cpu() - want).abs() / scale
        return e.max().item(), e.pow(2).mean().sqrt().item()
    mine_max, mine_rms = err(dgf.row_gemm(ad, dgf.packed_weight(wd, 0), K, N))
    torch.backends.cuda.matmul.allow_tf32 = False
    ref_max, ref_rms = err(ad @ wd.t())
    bf_max, bf_rms = err((ad.bfloat16() @ wd.bfloat16().t()).float())
    print(f"K={K} N={N} {data}: row_gemm max {mine_max:.2e} rms {mine_rms:.2e} | fp32 matmul max {ref_max:.2e} rms {ref_rms:.2e}"
          f" | bf16 matmul rms {bf_rms:.2e}")
    assert mine_rms < 1.5 * ref_rms and mine_max < 2.0 * ref_max
    assert (mine_rms < 1e-7 or data == "heavy_tailed") and bf_rms > 1e3 * mine_rms


@pytest.mark.parametrize("K,N", [(128, 128), (128, 384), (384, 128)])
def test_row_gemm_split_edge_rows(K, N):
    """Corners of the power-of-two scaling: all-zero rows (and one all-zero 128-chunk of a K = 384 row),
    rows of fp32 denormals, rows whose maximum is 2^120 (scale and inverse scale must stay inside the fp32 range),
    rows / weight columns containing inf or NaN.  Semantics of ``addmm`` in fp32: a zero row gives exactly the bias;
    finite rows are fp32-class accurate at any magnitude; a non-finite operand poisons exactly the outputs the
    reference poisons (its row, or its weight's output column) and nothing else.  One documented difference: a row
    containing +-inf yields NaN in that row (hi = inf, lo = inf - inf), where addmm yields +-inf or NaN."""
    from druggen_amd import functional as dgf
    R = 256
    a = _gen((R, K), 21).float()
    w = (_gen((N, K), 22) * 0.1).float()
    bias = _gen((N,), 23).float()
    a[3] = 0.0                                            # all-zero row
    a[4, :128] = 0.0                                      # one all-zero 128-chunk (K = 384: its own scale)
    a[5] = a[5] * 2.0 ** -140                             # fp32 denormals (|x| < 2^-126)
    a[6] = a[6] * 2.0 ** -126                             # straddles the normal / denormal boundary
    a[7] = a[7] / a[7].abs().max() * 2.0 ** 120           # maximum 2^120
    a[8, 5] = 3.0e38                                      # near FLT_MAX next to O(1) entries
    a[9] = 0.0
    a[9, 17] = 1.0                                        # a single non-zero
    finite_rows = list(range(3, 10)) + [0, 1, 2, 100, 255]
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    y = dgf.row_gemm(ad, dgf.packed_weight(wd, 0), K, N, bias=bd).cpu()
    want = a.double() @ w.double().t() + bias.double()
    scale = a.double().abs() @ w.double().abs().t() + bias.double().abs()
    assert torch.isfinite(y).all()
    assert torch.equal(y[3], bias)                                                 # zero row -> exactly the bias
    err = ((y.double() - want).abs() / scale.clamp_min(1e-300))[finite_rows]
    assert err.max() < 1e-6, err.max()                                             # fp32 class at every magnitude
    assert ((y[5].double() - want[5]).abs() <= 1e-6 * bias.double().abs() + 1e-37).all()   # denormal inputs: |error| ~ 0
    # non-finite activations: row 11 holds +inf, row 12 NaN, row 13 -inf and +inf
    a2 = a.clone()
    a2[11, 3] = float("inf")
    a2[12, 77 % K] = float("nan")
    a2[13, 0], a2[13, K - 1] = float("-inf"), float("inf")
    y2 = dgf.row_gemm(a2.cuda(), dgf.packed_weight(wd, 0), K, N, bias=bd).cpu()
    ref2 = (a2 @ w.t() + bias)
    assert torch.equal(torch.isfinite(y2), torch.isfinite(ref2))                    # the same rows, nothing else
    assert torch.equal(y2[torch.isfinite(ref2)], y[torch.isfinite(ref2)])           # other rows untouched, bit for bit
    assert torch.isnan(y2[12]).all() and not torch.isfinite(y2[11]).any() and not torch.isfinite(y2[13]).any()
    # non-finite weights: output column 2 sees an inf weight, column 5 a NaN weight
    w3 = w.clone()
    w3[2, 9] = float("inf")
    w3[5, 0] = float("nan")
    y3 = dgf.row_gemm(ad, dgf.packed_weight(w3.cuda(), 0), K, N, bias=bd).cpu()
    ref3 = (a @ w3.t() + bias)
    assert torch.equal(torch.isfinite(y3), torch.isfinite(ref3))
    keep = torch.isfinite(ref3)
    assert ((y3[keep].double() - y[keep].double()).abs() <= 1e-6 * scale[keep]).all()


@pytest.mark.parametrize("R", [5, 64, 1000, 2025 * 7])
def test_row_gemm_fused_epilogues(R):
    from druggen_amd import functional as dgf
    f = lambda t: t.float().cuda()
    # 384 -> 128 with bias + residual + LayerNorm (fc2 + ln6) and with residual only (fc1 dgrad)
    K, N = 384, 128
    a, w, b, res = _gen((R, K), 1), _gen((N, K), 3) * 0.1, _gen((N,), 4), _gen((R, N), 5)
    gamma, beta = 1 + 0.1 * _gen((N,), 6), _gen((N,), 7)
    pw = dgf.packed_weight(f(w), 0)
    y = dgf.row_gemm(f(a), pw, K, N, residual=f(res))
    assert _rel(y, a @ w.t() + res) < TOL
    y, mean, rstd, pre = dgf.row_gemm(f(a), pw, K, N, bias=f(b), residual=f(res), ln=(f(gamma), f(beta), 1e-5),
                                      want_pre=True)
    z = a @ w.t() + b + res
    y_ref, mu_ref, rs_ref = km.ln_fwd(z, gamma, beta)
    assert _rel(pre, z) < TOL
    assert _rel(y, y_ref) < TOL and _rel(mean, mu_ref.squeeze(-1)) < TOL and _rel(rstd, rs_ref.squeeze(-1)) < TOL
    # 128 -> 384 with bias + ReLU (fc1) and with an output mask (fc2 dgrad + ReLU backward)
    K, N = 128, 384
    a, w, b, om = _gen((R, K), 11), _gen((N, K), 12) * 0.1, _gen((N,), 13), _gen((R, N), 14)
    pw = dgf.packed_weight(f(w), 0)
    h, bits = dgf.row_gemm(f(a), pw, K, N, bias=f(b), relu=True, want_relu_bits=True)
    h_ref = torch.relu(a @ w.t() + b)
    assert _rel(h, h_ref) < TOL
    # the packed ReLU mask of that launch gates a later launch of the same geometry
    w3 = _gen((N, K), 15) * 0.1
    masked = dgf.row_gemm(f(a), dgf.packed_weight(f(w3), 0), K, N, mask_bits=bits)
    assert _rel(masked, (a @ w3.t()) * (h.double().cpu() > 0)) < TOL
    # 128 -> 128 with bias + residual + LayerNorm (out_e + ln4)
    K, N = 128, 128
    a, w, b, res = _gen((R, K), 21), _gen((N, K), 22) * 0.1, _gen((N,), 23), _gen((R, N), 24)
    y, mean, rstd = dgf.row_gemm(f(a), dgf.packed_weight(f(w), 0), K, N, bias=f(b), residual=f(res),
                                 ln=(f(gamma), f(beta), 1e-5))
    y_ref, _, _ = km.ln_fwd(a @ w.t() + b + res, gamma, beta)
    assert _rel(y, y_ref) < TOL


def test_packed_weight_cache_tracks_inplace_updates():
    from druggen_amd import functional as dgf
    w = torch.randn(128, 128, device="cuda")
    p1 = dgf.packed_weight(w, 0)
    assert dgf.packed_weight(w, 0) is p1
    w.add_(1.0)                                    # what an optimizer step does
    p2 = dgf.packed_weight(w, 0)
    assert p2 is not p1 and not torch.equal(p1, p2)


@pytest.mark.parametrize("hidden", ["f32", "dh16"])
@pytest.mark.parametrize("shape", [(2, 9, 9), (3, 50), (1, 45, 45)])
def test_fused_ffn_ln_matches_composite_all_orders(shape, hidden, hidden_mode):
    """dgf.linear_relu / dgf.linear_ln (first-order fast path and the create_graph
    fallback) against plain torch ops.  DG_HIDDEN=f32: float32-class throughout; dh16 (the default: dh and its second-order
    twin as one fp16 plane + row scales, h float32): the forward is untouched, gradients carry dh's 2^-11 rounding."""
    import torch.nn.functional as F
    from druggen_amd import functional as dgf
    hidden_mode(hidden)
    GT = 5 * TOL if hidden == "f32" else 5e-4
    C, H = 128, 384
    f = lambda t: t.float().cuda().requires_grad_(True)
    x = f(_gen(shape + (C,), 1))
    w1, b1 = f(_gen((H, C), 2) * 0.1), f(_gen((H,), 3))
    w2, b2 = f(_gen((C, H), 4) * 0.1), f(_gen((C,), 5))
    gamma, beta = f(1 + 0.1 * _gen((C,), 6)), f(_gen((C,), 7))
    dy = _gen(shape + (C,), 8).float().cuda()
    tx = _gen(shape + (C,), 9).float().cuda()
    params = [x, w1, b1, w2, b2, gamma, beta]

    def fused():
        return dgf.ffn_ln(x, w1, b1, w2, b2, gamma, beta, 1e-5)

    def plain():
        return F.layer_norm(x + F.linear(torch.relu(F.linear(x, w1, b1)), w2, b2), (C,), gamma, beta, 1e-5)

    y_f, y_p = fused(), plain()
    assert _rel(y_f, y_p.double().cpu()) < TOL
    g_f = torch.autograd.grad(y_f, params, dy)
    g_p = torch.autograd.grad(y_p, params, dy)
    for a, b in zip(g_f, g_p):
        assert _rel(a, b.double().cpu()) < GT
    # second order through the fused ops (reference loss.py usage without our context flag)
    gx_f = torch.autograd.grad(fused(), x, dy, create_graph=True)[0]
    gx_p = torch.autograd.grad(plain(), x, dy, create_graph=True)[0]
    s_f = torch.autograd.grad((gx_f * tx).sum(), [w1, w2, gamma])
    s_p = torch.autograd.grad((gx_p * tx).sum(), [w1, w2, gamma])
    for a, b in zip(s_f, s_p):
        assert _rel(a, b.double().cpu()) < GT
    # and with the flag (composite ops chosen at forward time)
    with dgf.second_order_forward():
        y_c = fused()
    gx_c = torch.autograd.grad(y_c, x, dy, create_graph=True)[0]
    s_c = torch.autograd.grad((gx_c * tx).sum(), [w1, w2, gamma])
    for a, b in zip(s_c, s_p):
        assert _rel(a, b.double().cpu()) < GT


def _ffn_f32_case(R, seed, scale=1.0):
    C, H = 128, 384
    f = lambda t: t.float().cuda().requires_grad_(True)
    return dict(x=f(_gen((R, C), seed + 1) * scale), w1=f(_gen((H, C), seed + 2) * 0.1), b1=f(_gen((H,), seed + 3)),
                w2=f(_gen((C, H), seed + 4) * 0.1), b2=f(_gen((C,), seed + 5)), gamma=f(1 + 0.1 * _gen((C,), seed + 6)),
                beta=f(_gen((C,), seed + 7)))


def _ffn_f32_masks(bits, R):
    """[R,384] bool from the mask words of dg_row_gemm's 128 -> 384 layout ([stage of 32 rows][8][64] words, bit
    (row block * 3 + channel block) * 4 + i): csrc/row_gemm_n384.hip."""
    st = (R + 31) // 32
    w = bits[:st * 512].view(st, 8, 4, 16).long().cpu()          # [stage, wave, channel quad, row in block]
    out = torch.zeros(st * 32, 384, dtype=torch.bool)
    for rb in range(2):
        for cb in range(3):
            for i in range(4):
                b = ((w >> ((rb * 3 + cb) * 4 + i)) & 1).bool()
                ch = 48 * torch.arange(8)[:, None] + 16 * cb + 4 * torch.arange(4)[None, :] + i            # [8,4]
                rows = 32 * torch.arange(st)[:, None] + 16 * rb + torch.arange(16)[None, :]                 # [st,16]
                out[rows[:, None, None, :].expand(st, 8, 4, 16), ch[None, :, :, None].expand(st, 8, 4, 16)] = b
    return out[:R]


@pytest.mark.parametrize("R,scale", [(1, 1.0), (17, 1.0), (129, 1.0), (1000, 1.0), (4097, 1e-9), (4097, 1e9), (33000, 1.0)])
def test_fused_float32_feed_forward_forward(R, scale):
    """dg_ffn_ln_fwd_f32 (csrc/ffn_fused_f32.hip: the [R,384] hidden tensor stays on chip) against float64: y, the pre-LayerNorm
    sum, the row statistics at 2e-5; the hi plane of h it leaves for the backward at the 2^-12 of one fp16 plane; the ReLU mask
    words equal to the sign of the float64 pre-activations wherever those are not within 1e-6 of zero; the gradients of the
    unchanged backward kernels on its outputs equal to the two-launch path's to rounding; tail tiles of every length."""
    from druggen_amd import functional as dgf
    C, H = 128, 384
    c = _ffn_f32_case(R, R % 11, scale)
    d = lambda t: t.detach().double().cpu()
    v64 = d(c["x"]) @ d(c["w1"]).t() + d(c["b1"])
    h64 = torch.relu(v64)
    pre64 = d(c["x"]) + h64 @ d(c["w2"]).t() + d(c["b2"])
    y64 = torch.nn.functional.layer_norm(pre64, (C,), d(c["gamma"]), d(c["beta"]), 1e-5)
    dy = _gen((R, C), 99).float().cuda()
    ins = [c[k] for k in ("x", "w1", "b1", "w2", "b2", "gamma", "beta")]
    got = {}
    try:
        for fused in (True, False):
            dgf.set_fused_ffn_f32(fused)
            y, pre, mean, rstd = dgf._FFNLN.apply(*ins, 1e-5)
            sv = y.grad_fn.saved_tensors
            got[fused] = dict(y=y.detach(), pre=pre.detach(), mean=mean, rstd=rstd, h=dgf.hidden_to_float(sv[7], R, H), bits=sv[11],
                              hbytes=sv[7].numel(), g=torch.autograd.grad(y, ins, dy))
    finally:
        dgf.set_fused_ffn_f32(True)
    a = got[True]
    assert a["hbytes"] == int(_lib().load().dg_hidden_bytes(R, H, _lib().F32_H16))      # one fp16 plane + row scales
    assert _rel(a["y"], y64) < TOL and _rel(a["pre"], pre64) < TOL
    assert _rel(a["mean"], pre64.mean(-1)) < TOL and _rel(a["rstd"], 1 / torch.sqrt(pre64.var(-1, unbiased=False) + 1e-5)) < TOL
    assert _rel(a["h"], h64) < 4e-4
    wrong = _ffn_f32_masks(a["bits"], R) != (v64 > 0)
    assert not wrong.any() or float((v64[wrong].abs() / v64.abs().max()).max()) < 1e-6
    # gradients of the unchanged backward kernels on both forwards.  A pre-activation within float32 rounding of zero may get
    # another mask bit in the two forwards (2 of 12.7 M elements at R = 33 000); such a row's dx differs by that unit's whole
    # contribution, so rows are compared where the two masks agree, and the parameter gradients with the flipped rows' share
    same = (_ffn_f32_masks(a["bits"], R) == _ffn_f32_masks(got[False]["bits"], R)).all(1)
    flips = int((~same).sum())
    assert flips <= max(2, R // 5000)
    assert _rel(a["g"][0][same.cuda()], got[False]["g"][0][same.cuda()].double().cpu()) < 2e-5
    for ga, gb in zip(a["g"][1:], got[False]["g"][1:]):
        assert _rel(ga, gb.double().cpu()) < (2e-5 if flips == 0 else 2e-3)


def test_fused_float32_feed_forward_degenerate_rows():
    """Rows and channels on which the kernel's power-of-two scales have nothing to hold on to: rows of x that are all zero, rows
    whose hidden units are ALL switched off (h = 0: the row scale of the stored plane clamps), a fc1 unit and a fc2 output with
    all-zero weights, one row 1e15 times larger than its neighbours -- y, the pre-LayerNorm sum, the stored plane and the
    gradients through the unchanged backward stay finite and equal to float64 / the two-launch path."""
    from druggen_amd import functional as dgf
    C, H, R = 128, 384, 333
    c = _ffn_f32_case(R, 7)
    with torch.no_grad():
        c["x"][::3] = 0.0                      # zero rows (their h = relu(b1): not zero)
        c["x"][1] *= 1e15                      # one huge row between ordinary ones (its variance, 1e30, still a float32)
        c["w1"][5] = 0.0                       # a hidden unit without weights
        c["w2"][9] = 0.0                       # an output channel without weights
        c["b1"][:] = c["b1"] - 0.2
    off = _ffn_f32_case(R, 8)
    with torch.no_grad():
        off["b1"][:] = -1e4                    # every hidden unit off in every row: h = 0, y = LN(x + b2)
    for case in (c, off):
        d = lambda t: t.detach().double().cpu()
        h64 = torch.relu(d(case["x"]) @ d(case["w1"]).t() + d(case["b1"]))
        pre64 = d(case["x"]) + h64 @ d(case["w2"]).t() + d(case["b2"])
        y64 = torch.nn.functional.layer_norm(pre64, (C,), d(case["gamma"]), d(case["beta"]), 1e-5)
        ins = [case[k] for k in ("x", "w1", "b1", "w2", "b2", "gamma", "beta")]
        dy = _gen((R, C), 98).float().cuda()
        got = {}
        try:
            for fused in (True, False):
                dgf.set_fused_ffn_f32(fused)
                y, pre, mean, rstd = dgf._FFNLN.apply(*ins, 1e-5)
                sv = y.grad_fn.saved_tensors
                got[fused] = dict(y=y.detach(), pre=pre.detach(), h=dgf.hidden_to_float(sv[7], R, H), g=torch.autograd.grad(y, ins, dy))
        finally:
            dgf.set_fused_ffn_f32(True)
        a, b = got[True], got[False]
        assert all(bool(torch.isfinite(t).all()) for t in (a["y"], a["pre"], a["h"]) + tuple(a["g"]))
        rows = torch.ones(R, dtype=torch.bool)
        rows[1] = False                        # (the 1e15 row dominates a whole-tensor norm: compared on its own)
        assert _rel(a["y"][rows.cuda()], y64[rows]) < TOL and _rel(a["y"][1], y64[1]) < TOL
        assert _rel(a["pre"][rows.cuda()], pre64[rows]) < TOL and _rel(a["pre"][1], pre64[1]) < TOL
        if float(h64.abs().max()) == 0.0:
            assert float(a["h"].abs().max()) == 0.0
        else:
            assert _rel(a["h"][rows.cuda()], h64[rows]) < 4e-4 and _rel(a["h"][1], h64[1]) < 4e-4
        for ga, gb in zip(a["g"], b["g"]):
            assert _rel(ga, gb.double().cpu()) < 2e-3


def test_fused_float32_feed_forward_node_rows_ride_and_launches_repeat():
    """The two-problem form (node rows ride in the launch over the edge rows, own weights) equals the two single launches bit
    for bit, without backward outputs too, and repeated launches are bit-identical (the kernel's counted vmcnt waits never let
    a weight fragment be read before it landed)."""
    from druggen_amd import functional as dgf
    cn, ce = _ffn_f32_case(180, 3), _ffn_f32_case(16200, 4)
    arg = lambda c: (c["w1"], c["b1"], c["w2"], c["b2"], c["gamma"], c["beta"], 1e-5)
    xo, yo, handle = dgf.ffn_ln_pair(cn["x"], arg(cn), ce["x"], arg(ce))
    assert handle is not None
    yn = dgf.ffn_ln(cn["x"], *arg(cn))
    ye = dgf.ffn_ln(ce["x"], *arg(ce))
    assert torch.equal(xo, yn) and torch.equal(yo, ye)
    with torch.no_grad():
        xo2, yo2, _ = dgf.ffn_ln_pair(cn["x"], arg(cn), ce["x"], arg(ce))
    assert torch.equal(xo2, xo) and torch.equal(yo2, yo)
    big = _ffn_f32_case(70001, 5)
    y0 = dgf.ffn_ln(big["x"], *arg(big)).detach().clone()
    for _ in range(20):
        assert torch.equal(dgf.ffn_ln(big["x"], *arg(big)), y0)


@pytest.mark.parametrize("shape", [(2, 9, 9), (7, 45)])
def test_fused_linear_ln_matches_composite_all_orders(shape):
    import torch.nn.functional as F
    from druggen_amd import functional as dgf
    C = 128
    f = lambda t: t.float().cuda().requires_grad_(True)
    x, res = f(_gen(shape + (C,), 1)), f(_gen(shape + (C,), 2))
    w, b = f(_gen((C, C), 3) * 0.1), f(_gen((C,), 4))
    gamma, beta = f(1 + 0.1 * _gen((C,), 6)), f(_gen((C,), 7))
    dy, tx = _gen(shape + (C,), 8).float().cuda(), _gen(shape + (C,), 9).float().cuda()
    params = [x, res, w, b, gamma, beta]
    fused = lambda: dgf.linear_ln(x, w, b, res, gamma, beta, 1e-5)
    plain = lambda: F.layer_norm(res + F.linear(x, w, b), (C,), gamma, beta, 1e-5)
    assert _rel(fused(), plain().double().cpu()) < TOL
    for a_, b_ in zip(torch.autograd.grad(fused(), params, dy), torch.autograd.grad(plain(), params, dy)):
        assert _rel(a_, b_.double().cpu()) < 5 * TOL
    gf = torch.autograd.grad(fused(), x, dy, create_graph=True)[0]
    gp = torch.autograd.grad(plain(), x, dy, create_graph=True)[0]
    for a_, b_ in zip(torch.autograd.grad((gf * tx).sum(), [w, gamma, res]),
                      torch.autograd.grad((gp * tx).sum(), [w, gamma, res])):
        assert _rel(a_, b_.double().cpu()) < 5 * TOL


@pytest.mark.parametrize("need_edge", [True, False])
def test_fused_attn_block_matches_reference_module_all_orders(need_edge):
    """dgf.attn_block (one autograd node) vs the oracle's mha + residual + LayerNorm, first order,
    create_graph fallback and second order."""
    import torch.nn.functional as F
    from druggen_amd import functional as dgf
    from druggen_amd.model.layers import MHA
    from oracle import druggen_oracle as orc
    torch.manual_seed(3)
    B, N, C, H = 2, 7, 128, 8
    attn = MHA(C, H).cuda()
    ln3, ln4 = torch.nn.LayerNorm(C).cuda(), torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        for ln in (ln3, ln4):
            ln.weight.add_(0.1 * torch.randn_like(ln.weight)); ln.bias.add_(0.1 * torch.randn_like(ln.bias))
    x1 = torch.randn(B, N, C, device="cuda", requires_grad=True)
    y = (0.5 * torch.randn(B, N, N, C, device="cuda")).requires_grad_(True)
    params = [p for p in attn.parameters()] + list(ln3.parameters()) + (list(ln4.parameters()) if need_edge else [])
    if not need_edge:
        params = [p for n_, p in attn.named_parameters() if not n_.startswith("out_e")] + list(ln3.parameters())
    P = {"attn." + k: v for k, v in attn.state_dict(keep_vars=True).items()}

    def ref():
        node_out, edge_out = orc.mha(P, "attn", x1, y, H)
        x2 = F.layer_norm(x1 + node_out, (C,), ln3.weight, ln3.bias, ln3.eps)
        y2 = F.layer_norm(y + edge_out, (C,), ln4.weight, ln4.bias, ln4.eps)
        return (x2, y2) if need_edge else (x2,)

    def mine():
        x2, y2 = dgf.attn_block(x1, y, attn, ln3, ln4, need_edge)
        return (x2, y2) if need_edge else (x2,)

    gouts = [torch.randn(B, N, C, device="cuda")] + ([torch.randn(B, N, N, C, device="cuda")] if need_edge else [])
    om, orf = mine(), ref()
    for a_, b_ in zip(om, orf):
        assert _rel(a_, b_.double().cpu()) < TOL
    gm = torch.autograd.grad(om, [x1, y] + params, gouts)
    gr = torch.autograd.grad(orf, [x1, y] + params, gouts)
    for a_, b_ in zip(gm, gr):
        assert _rel(a_, b_.double().cpu()) < 1e-4
    tx, ty = torch.randn_like(x1), torch.randn_like(y)
    sm = torch.autograd.grad(mine(), [x1, y], gouts, create_graph=True)
    sr = torch.autograd.grad(ref(), [x1, y], gouts, create_graph=True)
    for a_, b_ in zip(sm, sr):
        assert _rel(a_, b_.double().cpu()) < 1e-4
    hm = torch.autograd.grad((sm[0] * tx).sum() + (sm[1] * ty).sum(), [attn.q.weight, attn.e.weight, ln3.weight])
    hr = torch.autograd.grad((sr[0] * tx).sum() + (sr[1] * ty).sum(), [attn.q.weight, attn.e.weight, ln3.weight])
    for a_, b_ in zip(hm, hr):
        assert _rel(a_, b_.double().cpu()) < 2e-4


@pytest.mark.parametrize("act", ["relu", "leaky", "sigmoid", "tanh"])
@pytest.mark.parametrize("B,N,E", [(2, 6, 5), (3, 9, 3), (2, 45, 5), (1, 17, 10)])
def test_embed_sym_all_orders(act, B, N, E):
    """dg_embed_sym_fwd/bwd vs the oracle's embed() (Linear-act-Linear-act + symmetrise), incl. the
    gradient w.r.t. the input adjacency and the create_graph fallback."""
    from druggen_amd import functional as dgf
    from oracle import druggen_oracle as orc
    cfg = orc.NetConfig(act=act, vertexes=N, edges=E, nodes=4, dim=128)
    a = _gen((B, N, N, E), 1)
    w1, b1 = _gen((64, E), 2) * 0.5, _gen((64,), 3) * 0.3
    w2, b2 = _gen((128, 64), 4) * 0.2, _gen((128,), 5) * 0.3
    g = _gen((B, N, N, 128), 6)
    P = {"edge_layers.0.weight": w1, "edge_layers.0.bias": b1, "edge_layers.2.weight": w2, "edge_layers.2.bias": b2,
         "node_layers.0.weight": torch.zeros(64, 4, dtype=torch.float64), "node_layers.0.bias": torch.zeros(64, dtype=torch.float64),
         "node_layers.2.weight": torch.zeros(128, 64, dtype=torch.float64), "node_layers.2.bias": torch.zeros(128, dtype=torch.float64)}
    ins = [t.clone().requires_grad_(True) for t in (a, w1, b1, w2, b2)]
    Pr = dict(P, **{"edge_layers.0.weight": ins[1], "edge_layers.0.bias": ins[2], "edge_layers.2.weight": ins[3],
                    "edge_layers.2.bias": ins[4]})
    _, want = orc.embed(Pr, ins[0], torch.zeros(B, N, 4, dtype=torch.float64), cfg)
    gw = torch.autograd.grad(want, ins, g)
    f = lambda t: t.float().cuda().requires_grad_(True)
    dins = [f(t) for t in (a, w1, b1, w2, b2)]
    out = dgf.embed_sym(*dins, act)
    assert _rel(out, want.detach()) < TOL
    assert (out - out.permute(0, 2, 1, 3)).abs().max().item() == 0.0      # exactly symmetric
    gm = torch.autograd.grad(out, dins, g.float().cuda())
    for name, x, y in zip("da dw1 db1 dw2 db2".split(), gm, gw):
        assert _rel(x, y) < 1e-4, name
    # create_graph path (what the reference loss.py would trigger): second-order through the fallback
    ta = _gen((B, N, N, E), 7)
    gd = g.float().cuda().requires_grad_(True)
    g64 = g.clone().requires_grad_(True)
    ga = torch.autograd.grad(dgf.embed_sym(*dins, act), dins[0], gd, create_graph=True)[0]
    gr = torch.autograd.grad(orc.embed(Pr, ins[0], torch.zeros(B, N, 4, dtype=torch.float64), cfg)[1], ins[0], g64,
                             create_graph=True)[0]
    assert _rel(ga, gr.detach()) < 1e-4
    # second order (what the gradient penalty differentiates): adjoints of w1, w2 and of the upstream gradient.
    # relu / leaky run dg_embed_sym_bwd2 (act'' = 0); sigmoid / tanh the composite fallback
    h1 = torch.autograd.grad((ga * ta.float().cuda()).sum(), [dins[1], dins[3], gd])
    h2 = torch.autograd.grad((gr * ta).sum(), [ins[1], ins[3], g64])
    for name, x, y in zip("gw1 gw2 gg".split(), h1, h2):
        assert _rel(x, y) < 2e-4, name


def test_embed_second_order_entry_rejects_smooth_activations():
    """dg_embed_sym_bwd2 is the closed form for act'' = 0 only: sigmoid / tanh must be refused (the Python layer then
    uses the composite graph), never silently mis-differentiated."""
    lib = _lib().load()
    x = torch.zeros(64, device="cuda")
    args = [x.data_ptr()] * 11 + [x.data_ptr(), 1 << 30, 1, 9, 5, 64, 128]
    for act, want in ((2, -2), (3, -2)):
        st = lib.dg_embed_sym_bwd2(*args, act, 0, None)
        assert st == want and b"piecewise-linear" in lib.dg_last_error_string()


@pytest.mark.parametrize("act", ["relu", "leaky"])
@pytest.mark.parametrize("R", [1, 5, 64, 300, 512])
def test_discriminator_head_tail_all_orders(R, act):
    """dg_head_chain / dg_head_bwd / dg_head_wgrad (the Discriminator head after its first Linear, reference
    models.py:173-178) against torch.nn.Sequential in float64: output, first-order gradients of z1 and the six
    parameters, and the gradient-penalty pattern (gradient of a function of d out / d z1 with respect to the parameters
    and z1's upstream) -- the second order of the same node."""
    import torch.nn as nn
    from druggen_amd import functional as dgf
    torch.manual_seed(7 + R)
    mk = {"relu": nn.ReLU, "leaky": nn.LeakyReLU}[act]
    ref = nn.Sequential(mk(), nn.Linear(64, 32), mk(), nn.Linear(32, 16), mk(), nn.Linear(16, 1)).double()
    lay = [nn.Linear(64, 32), nn.Linear(32, 16), nn.Linear(16, 1)]
    for l, i in zip(lay, (1, 3, 5)):
        l.weight.data.copy_(ref[i].weight.data.float())
        l.bias.data.copy_(ref[i].bias.data.float())
        ref[i].weight.data.copy_(l.weight.data.double())
        ref[i].bias.data.copy_(l.bias.data.double())
        l.cuda()
    z = _gen((R, 64), 600 + R).float()
    w_out = _gen((R, 1), 601 + R).float()
    zc = z.cuda().requires_grad_(True)
    zd = z.double().requires_grad_(True)
    assert dgf.head_tail_supported(zc, lay, act)
    out = dgf.head_tail(zc, lay, act)
    outd = ref(zd)
    assert _rel(out.detach(), outd.detach()) < TOL
    params = [p for l in lay for p in (l.weight, l.bias)]
    paramsd = [p for i in (1, 3, 5) for p in (ref[i].weight, ref[i].bias)]
    got = torch.autograd.grad(out, [zc] + params, w_out.cuda(), retain_graph=True)
    want = torch.autograd.grad(outd, [zd] + paramsd, w_out.double(), retain_graph=True)
    for a_, b_ in zip(got, want):
        assert _rel(a_, b_) < TOL
    # gradient-penalty pattern: ((|d out / d z1| - 1)^2).mean() differentiated with respect to the parameters
    (gz,) = torch.autograd.grad(out, zc, torch.ones_like(out), create_graph=True)
    (gzd,) = torch.autograd.grad(outd, zd, torch.ones_like(outd), create_graph=True)
    pen = ((gz.norm(dim=1) - 1) ** 2).mean()
    pend = ((gzd.norm(dim=1) - 1) ** 2).mean()
    assert abs(pen.item() - pend.item()) < 1e-4 * max(1.0, abs(pend.item()))
    got2 = torch.autograd.grad(pen, [l.weight for l in lay])
    want2 = torch.autograd.grad(pend, [ref[i].weight for i in (1, 3, 5)])
    for a_, b_ in zip(got2, want2):
        assert _rel(a_, b_) < 5e-5
    if R % 2 == 0:
        # D(real) and D(fake) as one batch: upstream -1/B for the first half of the rows, +1/B for the second.  The reference
        # runs two passes whose last-bias gradients (-1, +1) cancel exactly; the summation by halves must too (AdamW turns a
        # residue of 1e-8 into a full-size step)
        B = R // 2
        gsign = torch.cat([torch.full((B, 1), -1.0 / B), torch.full((B, 1), 1.0 / B)]).cuda()
        gb = torch.autograd.grad(out, lay[2].bias, gsign, retain_graph=True)[0]
        assert gb.item() == 0.0
    # launches are bit-reproducible
    out2 = dgf.head_tail(zc, lay, act)
    got_again = torch.autograd.grad(out2, [zc] + params, w_out.cuda())
    assert torch.equal(out, out2) and all(torch.equal(a_, b_) for a_, b_ in zip(got, got_again))


@pytest.mark.parametrize("act", ["relu", "leaky"])
@pytest.mark.parametrize("B,N,E", [(1, 1, 5), (3, 9, 5), (4, 45, 13), (130, 45, 13), (2, 33, 16)])
def test_node_embedding_all_orders(B, N, E, act):
    """dg_embed_node_chain / dg_embed_node_bwd (node_layers, reference models.py:52-56, 154-158) against
    torch.nn.Sequential in float64: output, first-order gradients (input and the four parameters), and the
    gradient-penalty pattern (a function of d out / d z differentiated with respect to the parameters)."""
    import torch.nn as nn
    from druggen_amd import functional as dgf
    torch.manual_seed(11 + B + E)
    mk = {"relu": nn.ReLU, "leaky": nn.LeakyReLU}[act]
    l1, l2 = nn.Linear(E, 64), nn.Linear(64, 128)
    ref = nn.Sequential(nn.Linear(E, 64), mk(), nn.Linear(64, 128), mk()).double()
    for l, i in ((l1, 0), (l2, 2)):
        ref[i].weight.data.copy_(l.weight.data.double())
        ref[i].bias.data.copy_(l.bias.data.double())
        l.cuda()
    z = _gen((B, N, E), 700 + B).float()
    up = _gen((B, N, 128), 701 + B).float()
    zc, zd = z.cuda().requires_grad_(True), z.double().requires_grad_(True)
    assert dgf.node_embed_supported(zc, l1, l2, act)
    out, outd = dgf.node_embed(zc, l1, l2, act), ref(zd)
    assert out.shape == (B, N, 128) and _rel(out.detach(), outd.detach()) < TOL
    params = [l1.weight, l1.bias, l2.weight, l2.bias]
    paramsd = [ref[0].weight, ref[0].bias, ref[2].weight, ref[2].bias]
    got = torch.autograd.grad(out, [zc] + params, up.cuda(), retain_graph=True)
    want = torch.autograd.grad(outd, [zd] + paramsd, up.double(), retain_graph=True)
    for a_, b_ in zip(got, want):
        assert _rel(a_, b_) < TOL
    (gz,) = torch.autograd.grad(out, zc, up.cuda(), create_graph=True)
    (gzd,) = torch.autograd.grad(outd, zd, up.double(), create_graph=True)
    pen, pend = ((gz.norm(dim=-1) - 1) ** 2).mean(), ((gzd.norm(dim=-1) - 1) ** 2).mean()
    assert abs(pen.item() - pend.item()) < 1e-4 * max(1.0, abs(pend.item()))
    got2 = torch.autograd.grad(pen, [l1.weight, l2.weight])
    want2 = torch.autograd.grad(pend, [ref[0].weight, ref[2].weight])
    for a_, b_ in zip(got2, want2):
        assert _rel(a_, b_) < 5e-5
    out2 = dgf.node_embed(zc, l1, l2, act)
    assert torch.equal(out, out2)
    lib = _lib().load()
    assert lib.dg_embed_node_chain(zc.data_ptr(), None, None, l1.weight.data_ptr(), None, l2.weight.data_ptr(), None,
                                   out.data_ptr(), out.data_ptr(), 1, 17, 0, torch.cuda.current_stream().cuda_stream) != 0


# ---------------------------------------------------------------------------------------------------------------------------
# DG_DTYPE_F32_H16: the [R,384] hidden tensors of the float32 feed-forward as one fp16 plane + one inverse scale per row
# ---------------------------------------------------------------------------------------------------------------------------
def _h16_chain(R, seed=300, row_scales=None):
    from druggen_amd import _lib as L, functional as dgf
    C, H = 128, 384
    x = _gen((R, C), seed).float().cuda()
    if row_scales is not None:
        x = x * row_scales[:, None].cuda()
    w1 = (_gen((H, C), seed + 1) * 0.1).float().cuda()
    b1 = (_gen((H,), seed + 2) * (0.0 if row_scales is not None else 1.0)).float().cuda()
    w2 = (_gen((C, H), seed + 3) * 0.1).float().cuda()
    b2 = _gen((C,), seed + 4).float().cuda()
    g, be = (_gen((C,), seed + 5) * 0.1 + 1).float().cuda(), _gen((C,), seed + 6).float().cuda()
    dz = (_gen((R, C), seed + 7) * 1e-3).float().cuda()
    pw = lambda w, m: dgf.packed_weight(w, m, torch.float32)
    return L, dgf, dict(x=x, w1=w1, b1=b1, w2=w2, b2=b2, g=g, be=be, dz=dz, pw=pw, C=C, H=H)


def _hidden_bound_doc():
    """f32s ("split", DG_DTYPE_F32_H32): hi + lo fp16 planes under one row scale: 22 significand bits of every element within
    2^-18 of its row maximum -- bound 2^-21 relative + 2^-39 of the row maximum."""


def _hidden_bound(ref, fmt):
    """Per-element bound of a narrow hidden storage against the float32 kernel's value: fp16 plane -- half an fp16 ulp (2^-11
    relative) + the denormal floor 2^-25 of the row maximum; three-byte elements -- half a unit of the 16th significant bit."""
    if fmt == "f24":
        return ref.abs() * 2.0 ** -16
    if fmt == "f32s":
        return ref.abs() * 2.0 ** -21 + ref.abs().amax(1, keepdim=True) * 2.0 ** -39
    return ref.abs() * 2.0 ** -11 + ref.abs().amax(1, keepdim=True) * 2.0 ** -25


def _hidden_code(L, fmt):
    return {"f24": L.F32_H24, "f32s": L.F32_H32}.get(fmt, L.F32_H16)


@pytest.mark.parametrize("fmt", ["f24", "f16", "f32s"])
@pytest.mark.parametrize("R", [1, 15, 16, 17, 33, 1000, 4097, 70000])
def test_hidden_fp16_plane_writer_reader_and_weight_gradients(R, fmt):
    """128 -> 384 row GEMM writing DG_DTYPE_F32_H24 / _H16, 384 -> 128 row GEMM and both weight-gradient shapes reading it
    (reference layers.py:50-53 forward / backward).  (a) every decoded element is within the storage's rounding of the
    float32 kernel's value (``_hidden_bound``) and the ReLU bit masks are identical; (b) GIVEN the decoded operand, the
    readers are float32-class: fp64 over the decoded values at TOL."""
    L, dgf, t = _h16_chain(R)
    CODE = _hidden_code(L, fmt)
    x, dz, pw, C, H = t["x"], t["dz"], t["pw"], t["C"], t["H"]
    h32, bits32 = dgf.row_gemm(x, pw(t["w1"], 0), C, H, bias=t["b1"], relu=True, want_relu_bits=True)
    h16, bits16 = dgf.row_gemm(x, pw(t["w1"], 0), C, H, bias=t["b1"], relu=True, want_relu_bits=True, code=CODE)
    nw = (R + 31) // 32 * 512
    assert torch.equal(bits32[:nw], bits16[:nw])
    hd = dgf.hidden_to_float(h16, R)
    assert bool(((hd - h32).abs() <= _hidden_bound(h32, fmt)).all())
    hd64 = hd.double().cpu()
    # readers: the four 384 -> 128 epilogues
    for res, ln in ((False, False), (True, False), (False, True), (True, True)):
        out = dgf.row_gemm(h16, pw(t["w2"], 0), H, C, bias=t["b2"], residual=x if res else None,
                           ln=(t["g"], t["be"], 1e-5) if ln else None, want_pre=ln, R=R)
        want = hd64 @ t["w2"].double().cpu().t() + t["b2"].double().cpu() + (x.double().cpu() if res else 0.0)
        if ln:
            y, mean, rstd, pre = out
            assert _rel(pre, want) < TOL
            assert _rel(y, torch.nn.functional.layer_norm(want, (C,), t["g"].double().cpu(), t["be"].double().cpu(), 1e-5)) < TOL
            assert _rel(mean, want.mean(1)) < TOL
        else:
            assert _rel(out, want) < TOL
    # mask-in writer + dx reader
    dh16 = dgf.row_gemm(dz, pw(t["w2"], 1), C, H, mask_bits=bits16, code=CODE)
    dh32 = dgf.row_gemm(dz, pw(t["w2"], 1), C, H, mask_bits=bits32)
    dhd = dgf.hidden_to_float(dh16, R)
    if fmt == "f16":      # the backward's fp16-plane writer takes its activation rows as ONE fp16 plane too (two products: the test below)
        assert _rel(dhd, dh32.double().cpu()) < 3.5e-4
    else:
        assert bool(((dhd - dh32).abs() <= _hidden_bound(dh32, fmt)).all())
    dx = dgf.row_gemm(dh16, pw(t["w1"], 1), H, C, residual=dz, R=R)
    assert _rel(dx, dz.double().cpu() + dhd.double().cpu() @ t["w1"].double().cpu()) < TOL
    # weight gradients: dW2 = dz^T h (x operand hidden), dW1 = dh^T x (dy operand hidden), with their bias sums
    dw2, db2 = dgf._wgrad(dz, h16, True)
    if fmt == "f32s":      # (the weight gradient reads the hi plane of a pre-split tensor only)
        off = int(L.load().dg_hidden_scale_offset(R, H))
        hi64 = (h16[:R * H * 2].view(torch.float16).view(R, H).double() * h16[2 * off:2 * off + 4 * R].view(torch.float32).double()[:, None]).cpu()
        assert _rel(dw2, dz.double().cpu().t() @ hi64) < TOL and _rel(dw2, dz.double().cpu().t() @ hd64) < 5e-4
    else:
        assert _rel(dw2, dz.double().cpu().t() @ hd64) < TOL
    assert _rel(db2, dz.double().cpu().sum(0)) < TOL
    dw1, db1 = dgf._wgrad(dh16, x, True)
    if fmt == "f32s":
        off = int(L.load().dg_hidden_scale_offset(R, H))
        dhi64 = (dh16[:R * H * 2].view(torch.float16).view(R, H).double() * dh16[2 * off:2 * off + 4 * R].view(torch.float32).double()[:, None]).cpu()
        assert _rel(dw1, dhi64.t() @ x.double().cpu()) < TOL and _rel(db1, dhi64.sum(0)) < TOL
    else:
        assert _rel(dw1, dhd.double().cpu().t() @ x.double().cpu()) < TOL and _rel(db1, dhd.double().cpu().sum(0)) < TOL
    # bit-reproducible
    dw2b, _ = dgf._wgrad(dz, h16, True)
    assert torch.equal(dw2, dw2b)


@pytest.mark.parametrize("R", [17, 1000, 70000])
def test_backward_fp16_plane_two_product_arithmetic(R):
    """The arithmetic of the BACKWARD's fp16-plane writer: dh = (dz W2) * m with the activation rows as ONE fp16 plane (products
    w_hi.x_hi + w_lo.x_hi; the result carries a 2^-11 rounding anyway).  Against fp64: relative L2 error of dh below 3.5e-4
    (2.1e-4 with all three products; a single product measured 1.15e-3 on a golden and is not offered); the ReLU mask still
    zeroes exactly, and dx = dz + dh W1 on the stored plane is float32 class."""
    L, dgf, t = _h16_chain(R)
    x, dz, pw, C, H = t["x"], t["dz"], t["pw"], t["C"], t["H"]
    _, bits = dgf.row_gemm(x, pw(t["w1"], 0), C, H, bias=t["b1"], relu=True, want_relu_bits=True)
    h64 = torch.relu(x.double().cpu() @ t["w1"].double().cpu().t() + t["b1"].double().cpu())
    dh64 = (dz.double().cpu() @ t["w2"].double().cpu()) * (h64 > 0)
    dh = dgf.row_gemm(dz, pw(t["w2"], 1), C, H, mask_bits=bits, code=L.F32_H16)
    dhd = dgf.hidden_to_float(dh, R)
    # (mask flips between the float32 kernel's mask and fp64's are excluded: compare where both agree)
    keep = ((dhd != 0).cpu() == (dh64 != 0))
    e_dh = ((dhd.double().cpu() - dh64) * keep).norm() / dh64.norm()
    assert float(e_dh) < 3.5e-4, float(e_dh)
    assert bool(((dhd == 0) | (dh64.cuda() != 0) | ~keep.cuda()).all())
    dx = dgf.row_gemm(dh, pw(t["w1"], 1), H, C, residual=dz, R=R)
    e_dx = (dx.double().cpu() - (dz.double().cpu() + dhd.double().cpu() @ t["w1"].double().cpu())).norm() / dx.double().norm().cpu()
    assert float(e_dx) < TOL, float(e_dx)


@pytest.mark.parametrize("fmt", ["f24", "f16", "f32s"])
def test_hidden_fp16_plane_row_scales_cover_the_float32_range(fmt):
    """One power-of-two scale per ROW (fp16 plane) / the float32 exponent of every element (three-byte elements): rows 2^+-60
    apart, an all-zero row and a row whose values span 2^20 keep the per-element bound; the 384 -> 128 reader un-scales
    exactly."""
    R = 64
    scales = torch.tensor([2.0 ** ((i * 7) % 121 - 60) for i in range(R)], dtype=torch.float64)
    scales[5] = 0.0
    L, dgf, t = _h16_chain(R, seed=340, row_scales=scales.float())
    x, pw, C, H = t["x"], t["pw"], t["C"], t["H"]
    x[9] = x[9] * torch.logspace(0, 6, C, base=10.0, device="cuda")       # a heavy-tailed row
    h32 = dgf.row_gemm(x, pw(t["w1"], 0), C, H)
    h16 = dgf.row_gemm(x, pw(t["w1"], 0), C, H, code=_hidden_code(L, fmt))
    hd = dgf.hidden_to_float(h16, R)
    assert bool(torch.isfinite(hd).all()) and float(hd[5].abs().max()) == 0.0
    assert bool(((hd - h32).abs() <= _hidden_bound(h32, fmt)).all())
    y = dgf.row_gemm(h16, pw(t["w2"], 0), H, C, R=R)
    want = hd.double().cpu() @ t["w2"].double().cpu().t()
    err = (y.double().cpu() - want).norm(dim=1) / want.norm(dim=1).clamp_min(1e-300)
    assert float(err.max()) < TOL


@pytest.mark.parametrize("fmt", ["f24", "f16", "f32s"])
@pytest.mark.parametrize("Rn,Re", [(360, 70000), (33, 66000)])
def test_hidden_fp16_plane_riding_launches_equal_separate_launches(Rn, Re, fmt, monkeypatch):
    """Riding launches (pair.h) with DG_DTYPE_F32_H16 operands: row GEMMs bit-identical to separate launches, weight
    gradients equal to rounding; a float32 rider is never paired with an fp16-plane carrier (launched on its own, first)."""
    from druggen_amd import _lib as L, functional as dgf
    ts = [_h16_chain(R, seed=360 + 20 * i)[2] for i, R in enumerate((Rn, Re))]
    Rs = (Rn, Re)
    CODE = _hidden_code(L, fmt)

    def chain(paired):
        with dgf._pair_launches(ts[0]["x"], on=paired):
            hs = [dgf.row_gemm(t["x"], t["pw"](t["w1"], 0), 128, 384, bias=t["b1"], relu=True, want_relu_bits=True, code=CODE) for t in ts]
            ys = [dgf.row_gemm(h, t["pw"](t["w2"], 0), 384, 128, bias=t["b2"], residual=t["x"], ln=(t["g"], t["be"], 1e-5), want_pre=True,
                               R=R) for (h, _), t, R in zip(hs, ts, Rs)]
            dh = [dgf.row_gemm(t["dz"], t["pw"](t["w2"], 1), 128, 384, mask_bits=b, code=CODE) for (_, b), t in zip(hs, ts)]
            dx = [dgf.row_gemm(d, t["pw"](t["w1"], 1), 384, 128, residual=t["dz"], R=R) for d, t, R in zip(dh, ts, Rs)]
            wg = dgf._wgrad_many([(ts[0]["dz"], hs[0][0], True), (ts[1]["dz"], hs[1][0], True), (dh[0], ts[0]["x"], True), (dh[1], ts[1]["x"], True)])
        return [[hs[i][0], *ys[i], dh[i], dx[i]] for i in range(2)], wg

    sep, wsep = chain(False)
    par, wpar = chain(True)
    for a, b in zip(sep, par):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    for (dw, db), (ew, eb) in zip(wsep, wpar):
        assert _rel(ew, dw.double().cpu()) < 2e-6 and _rel(eb, db.double().cpu()) < 2e-6
    # mixed storage inside one region: the float32 rider leaves on its own before the fp16-plane launch
    with dgf._pair_launches(ts[0]["x"]):
        hn = dgf.row_gemm(ts[0]["x"], ts[0]["pw"](ts[0]["w1"], 0), 128, 384, bias=ts[0]["b1"])
        he = dgf.row_gemm(ts[1]["x"], ts[1]["pw"](ts[1]["w1"], 0), 128, 384, bias=ts[1]["b1"], code=CODE)
    assert torch.equal(hn, dgf.row_gemm(ts[0]["x"], ts[0]["pw"](ts[0]["w1"], 0), 128, 384, bias=ts[0]["b1"]))
    assert torch.equal(he, dgf.row_gemm(ts[1]["x"], ts[1]["pw"](ts[1]["w1"], 0), 128, 384, bias=ts[1]["b1"], code=CODE))


def test_hidden_split_planes_edge_rows():
    """Corners of the row scale of DG_DTYPE_F32_H32 (the forward's pre-split hidden tensor) and of the fp16-plane storage:
    all-zero rows, rows of denormals, rows whose maximum is 2^120 or sits next to O(1) entries, a single non-zero; rows holding
    inf / NaN poison exactly their own row in the writer and in the 384 -> 128 reader, every other row is untouched bit for bit."""
    from druggen_amd import _lib as L, functional as dgf
    R, C, H = 256, 128, 384
    a = _gen((R, C), 21).float()
    w1 = (_gen((H, C), 22) * 0.1).float().cuda()
    w2 = (_gen((C, H), 23) * 0.1).float().cuda()
    a[3] = 0.0
    a[5] = a[5] * 2.0 ** -140
    a[6] = a[6] * 2.0 ** -126
    a[7] = a[7] / a[7].abs().max() * 2.0 ** 100
    a[8, 5] = 1.0e30
    a[9] = 0.0
    a[9, 17] = 1.0
    pw = lambda w, m: dgf.packed_weight(w, m, torch.float32)
    h32 = dgf.row_gemm(a.cuda(), pw(w1, 0), C, H)
    for fmt in ("f32s", "f16"):
        code = _hidden_code(L, fmt)
        hb = dgf.row_gemm(a.cuda(), pw(w1, 0), C, H, code=code)
        hd = dgf.hidden_to_float(hb, R)
        assert bool(torch.isfinite(hd).all()) and float(hd[3].abs().max()) == 0.0
        assert bool(((hd - h32).abs() <= _hidden_bound(h32, fmt)).all()), fmt
        y = dgf.row_gemm(hb, pw(w2, 0), H, C, R=R)
        want = hd.double().cpu() @ w2.double().cpu().t()
        scale = hd.double().cpu().abs() @ w2.double().cpu().abs().t()
        assert bool(torch.isfinite(y).all())
        # (float32 class at every magnitude; results that are float32 denormals themselves -- row 5 -- carry an absolute floor)
        assert bool(((y.double().cpu() - want).abs() <= 1e-6 * scale + 1e-37).all()), fmt
        # non-finite rows
        a2 = a.clone()
        a2[11, 3] = float("inf")
        a2[12, 77] = float("nan")
        hb2 = dgf.row_gemm(a2.cuda(), pw(w1, 0), C, H, code=code)
        hd2 = dgf.hidden_to_float(hb2, R)
        good = torch.ones(R, dtype=torch.bool)
        good[11] = good[12] = False
        assert not bool(torch.isfinite(hd2[11]).any()) and not bool(torch.isfinite(hd2[12]).any())
        assert torch.equal(hd2[good.cuda()], hd[good.cuda()])
        y2 = dgf.row_gemm(hb2, pw(w2, 0), H, C, R=R)
        assert not bool(torch.isfinite(y2[11]).any()) and not bool(torch.isfinite(y2[12]).any())
        assert torch.equal(y2[good.cuda()], y[good.cuda()])
