#!/usr/bin/env python
"""Micro-benchmarks of individual C-ABI kernels at configs[1] shapes (GPU box).
Usage: python scripts/bench_kernels.py [wgrad|attn|ln|gemm|embed|all] [f32|bf16] [batch=256]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import functional as dgf  # noqa: E402


DT = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ES = 2 if DT == torch.bfloat16 else 4


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)   # let the host run ahead: the events then time the GPU, not the launch path
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def wgrad():
    R = BATCH * 45 * 45
    for (N, K) in [(128, 128), (384, 128), (128, 384), (128, 64)]:
        dy = torch.randn(R, N, device="cuda").to(DT)
        x = torch.randn(R, K, device="cuda").to(DT)
        t_mine = timeit(lambda: dgf._wgrad(dy, x, True))
        t_lib = timeit(lambda: (dy.t().mm(x), dy.sum(0)))
        t_mm = timeit(lambda: dy.t().mm(x))
        fl = 2.0 * R * N * K
        by = float(ES) * R * (N + K)
        print(f"wgrad R={R} N={N} K={K}: mine {t_mine:8.1f} us ({fl / t_mine / 1e6:6.1f} TF, {by / t_mine / 1e3:6.0f} GB/s)"
              f" | lib mm+sum {t_lib:8.1f} us  mm only {t_mm:8.1f} us ({fl / t_mm / 1e6:6.1f} TF)")
    R = 256 * 45
    dy = torch.randn(R, 128, device="cuda"); x = torch.randn(R, 128, device="cuda")
    print(f"wgrad node rows R={R} 128x128: mine {timeit(lambda: dgf._wgrad(dy, x, True)):.1f} us | lib {timeit(lambda: (dy.t().mm(x), dy.sum(0))):.1f} us")


def attn():
    B, N, C = 256, 45, 128
    q, k, v = (torch.randn(B, N, C, device="cuda") for _ in range(3))
    e = torch.randn(B, N, N, C, device="cuda") * 0.5
    ws, wo = torch.randn(B, N, N, C, device="cuda"), torch.randn(B, N, C, device="cuda")
    t = timeit(lambda: dgf._AttnCore.apply(q, k, v, e, 0.25, True))
    print(f"attn fwd: {t:.1f} us  {4 * B * (2 * N * N * C + 4 * N * C) / t / 1e3:.0f} GB/s")
    t = timeit(lambda: dgf._AttnCoreBwd.apply(q, k, v, e, ws, wo, 0.25))
    print(f"attn bwd: {t:.1f} us  {4 * B * (3 * N * N * C + 7 * N * C) / t / 1e3:.0f} GB/s")
    tq = [torch.randn(B, N, C, device="cuda") for _ in range(3)] + [torch.randn(B, N, N, C, device="cuda")]
    qq = [x.requires_grad_(True) for x in (q, k, v, e, ws, wo)]
    def bwd2():
        g = dgf._AttnCoreBwd.apply(*qq, 0.25)
        torch.autograd.grad(g, qq, tq)
    tb = timeit(bwd2)
    print(f"attn bwd+bwd2: {tb:.1f} us")


def ln():
    R, C = BATCH * 45 * 45, 128
    a, r = torch.randn(R, C, device="cuda").to(DT), torch.randn(R, C, device="cuda").to(DT)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    t = timeit(lambda: dgf.ln_residual(a, r, g, b))
    print(f"ln fwd: {t:.1f} us {ES * R * C * 3 / t / 1e3:.0f} GB/s")
    t2 = timeit(lambda: torch.nn.functional.layer_norm(a + r, (C,), g.to(DT), b.to(DT)))
    print(f"torch add+ln fwd: {t2:.1f} us")
    pre = a + r
    mean, rstd = pre.float().mean(1), (pre.float().var(1, unbiased=False) + 1e-5).rsqrt()
    dy = torch.randn(R, C, device="cuda").to(DT)
    t = timeit(lambda: dgf._ln_bwd_rows(pre, g, mean, rstd, dy))
    print(f"ln bwd (+finish): {t:.1f} us {ES * R * C * 3 / t / 1e3:.0f} GB/s")
    t3 = timeit(lambda: a.clone())
    print(f"torch copy 265MB: {t3:.1f} us  {2 * 4 * R * C / t3 / 1e3:.0f} GB/s")


def gemm():
    R = BATCH * 45 * 45
    for (K, N) in [(128, 128), (128, 384), (384, 128)]:
        a = torch.randn(R, K, device="cuda").to(DT)
        w = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        pw = dgf.packed_weight(w, 0, DT)
        wl, bl = w.to(DT), b.to(DT)
        t_mine = timeit(lambda: dgf.row_gemm(a, pw, K, N, bias=b))
        t_lib = timeit(lambda: torch.nn.functional.linear(a, wl, bl))
        fl = 2.0 * R * N * K
        gb = ES * R * (K + N) / 1e3
        print(f"row_gemm fwd K={K} N={N}: mine {t_mine:8.1f} us ({fl / t_mine / 1e6:6.1f} TF, {gb / t_mine:6.0f} GB/s) | F.linear {t_lib:8.1f} us ({fl / t_lib / 1e6:6.1f} TF)")
        pwt = dgf.packed_weight(torch.randn(K, N, device="cuda") * 0.05, 1, DT)
        t_d = timeit(lambda: dgf.row_gemm(a, pwt, K, N))
        print(f"   dgrad-mode (same shape): mine {t_d:8.1f} us ({fl / t_d / 1e6:6.1f} TF)")
        if N == 384:
            t_mine = timeit(lambda: dgf.row_gemm(a, pw, K, N, bias=b, relu=True))
            t_lib = timeit(lambda: torch.relu(torch.nn.functional.linear(a, wl, bl)))
            print(f"   + relu epilogue: mine {t_mine:8.1f} us | lib linear+relu {t_lib:8.1f} us")
        if N == 128:
            res = torch.randn(R, N, device="cuda").to(DT); g = torch.ones(N, device="cuda"); be = torch.zeros(N, device="cuda")
            t_mine = timeit(lambda: dgf.row_gemm(a, pw, K, N, bias=b, residual=res, ln=(g, be, 1e-5)))
            t_lib = timeit(lambda: torch.nn.functional.layer_norm(res + torch.nn.functional.linear(a, wl, bl), (N,), g.to(DT), be.to(DT)))
            print(f"   + residual+LN epilogue: mine {t_mine:8.1f} us | lib linear+add+LN {t_lib:8.1f} us")


def embed():
    B, N, E = BATCH, 45, 5
    a = torch.zeros(B, N, N, E, device="cuda"); a[..., 0] = 1
    w1, b1 = torch.randn(64, E, device="cuda") * 0.3, torch.randn(64, device="cuda") * 0.1
    w2, b2 = torch.randn(128, 64, device="cuda") * 0.1, torch.randn(128, device="cuda") * 0.1
    t = timeit(lambda: dgf._EmbedSym.apply(a, w1, b1, w2, b2, "relu", DT))
    print(f"embed_sym fwd: {t:.1f} us")
    t = timeit(lambda: dgf._composite_embed_sym(a, w1, b1, w2, b2, "relu"))
    print(f"composite fwd: {t:.1f} us")
    g = torch.randn(B, N, N, 128, device="cuda").to(DT)
    for need_da in (False, True):
        ins = [a.clone().requires_grad_(need_da)] + [x.clone().requires_grad_(True) for x in (w1, b1, w2, b2)]
        def fb():
            out = dgf._EmbedSym.apply(*ins, "relu", DT)
            torch.autograd.grad(out, [i for i in ins if i.requires_grad], g)
        def fbc():
            out = dgf._composite_embed_sym(*ins, "relu")
            torch.autograd.grad(out, [i for i in ins if i.requires_grad], g)
        print(f"embed_sym fwd+bwd (da={need_da}): {timeit(fb):.1f} us | composite {timeit(fbc):.1f} us")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    for name, fn in (("wgrad", wgrad), ("attn", attn), ("ln", ln), ("gemm", gemm), ("embed", embed)):
        if which in (name, "all"):
            fn()
