#!/usr/bin/env python
"""A/B timing of row_gemm K=384->128 across debug builds (developer tool): run once per DG_LIB."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import functional as dgf
from bench_kernels import timeit
R = 256 * 45 * 45
for (K, N) in [(384, 128), (128, 384), (128, 128)]:
    a = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    pw = dgf.packed_weight(w, 0)
    t = timeit(lambda: dgf.row_gemm(a, pw, K, N, bias=b))
    line = f"{os.environ.get('DG_LIB', 'default')[-10:]} K={K} N={N}: plain {t:7.1f} us"
    if N == 128:
        res = torch.randn(R, N, device="cuda"); g = torch.ones(N, device="cuda"); be = torch.zeros(N, device="cuda")
        t2 = timeit(lambda: dgf.row_gemm(a, pw, K, N, bias=b, residual=res, ln=(g, be, 1e-5)))
        line += f"  +res+LN {t2:7.1f} us"
    print(line, flush=True)
