#!/usr/bin/env python
"""Node-level row GEMMs (R = B*N rows): time per launch for the shapes of the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from druggen_amd import functional as dgf

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 45
torch.manual_seed(0)
def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n * 1e3
for K, N in ((128, 128), (128, 384), (384, 128)):
    a = torch.randn(R, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.1
    b = torch.randn(N, device="cuda")
    res = torch.randn(R, N, device="cuda")
    g, be = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
    pk = dgf.packed_weight(w, 0)
    t0 = timeit(lambda: dgf.row_gemm(a, pk, K, N, bias=b))
    t1 = timeit(lambda: dgf.row_gemm(a, pk, K, N, bias=b, residual=res, ln=(g, be, 1e-5), want_pre=True)) if N == 128 else 0.0
    print(f"R={R} {K}->{N}: plain {t0:6.1f} us   +res+LN {t1:6.1f} us")
x = torch.randn(R, 128, device="cuda"); dy = torch.randn(R, 128, device="cuda")
print(f"wgrad 128x128 R={R}: {timeit(lambda: dgf._wgrad(dy, x, True)):6.1f} us")
pre = torch.randn(R, 128, device="cuda"); gm = torch.rand(128, device="cuda") + 0.5
mean, rstd = pre.mean(-1), (pre.var(-1, unbiased=False) + 1e-5).rsqrt()
print(f"ln_bwd R={R}: {timeit(lambda: dgf._ln_bwd_rows(pre, gm, mean, rstd, dy)):6.1f} us")
print(f"empty launch floor: {timeit(lambda: torch.empty(1, device='cuda').zero_()):6.1f} us")
