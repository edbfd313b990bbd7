#!/usr/bin/env python
"""Input-gradient GEMM + LayerNorm backward at configs[1] size (R = 518 400, fp32): the fused epilogue launch against
the two launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from druggen_amd import functional as dgf

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 45 * 45
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n * 1e3
for K in (128,):
    a = torch.randn(R, K, device="cuda") * 1e-2
    w = torch.randn(K, 128, device="cuda") * 0.1
    res = torch.randn(R, 128, device="cuda") * 1e-2
    pre = torch.randn(R, 128, device="cuda")
    gamma = torch.rand(128, device="cuda") + 0.5
    mean, rstd = pre.mean(-1), (pre.var(-1, unbiased=False) + 1e-5).rsqrt()
    packed = dgf.packed_weight(w, 1)
    fused = timeit(lambda: dgf.row_gemm_ln_bwd(a, packed, K, res, pre, gamma, mean, rstd))
    gemm = timeit(lambda: dgf.row_gemm(a, packed, K, 128, residual=res))
    dy = dgf.row_gemm(a, packed, K, 128, residual=res)
    ln = timeit(lambda: dgf._ln_bwd_rows(pre, gamma, mean, rstd, dy))
    print(f"K={K} R={R}: fused {fused:7.1f} us | gemm {gemm:7.1f} + ln_bwd {ln:7.1f} = {gemm + ln:7.1f} us")
