#!/usr/bin/env python
"""Fused float32 attention-half forward (dg_attn_half_f32_fwd) against the three launches it replaces (e projection, attention
core, out_e + residual + LayerNorm) at BASELINE configs[1] shapes (developer tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import functional as dgf, _lib
from bench_kernels import timeit
B, N, C = int(os.environ.get("B", 256)), 45, 128
dev = "cuda"
torch.manual_seed(0)
y = torch.randn(B, N, N, C, device=dev)
q, k, v = (torch.randn(B, N, C, device=dev) for _ in range(3))
we, woe = torch.randn(C, C, device=dev) * 0.1, torch.randn(C, C, device=dev) * 0.1
be, boe = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
g4, b4 = torch.randn(C, device=dev) * 0.1 + 1, torch.randn(C, device=dev) * 0.1
lib = _lib.load()
R = B * N * N
yf = y.view(R, C)
e, s, y2, pre = (torch.empty(R, C, device=dev) for _ in range(4))
o = torch.empty(B, N, C, device=dev)
mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
pe, po = dgf.packed_weight(we, 0), dgf.packed_weight(woe, 0)
def fused(keep):
    _lib.check(lib.dg_attn_half_f32_fwd(y.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), pe.data_ptr(), be.data_ptr(),
                                        po.data_ptr(), boe.data_ptr(), g4.data_ptr(), b4.data_ptr(), e.data_ptr() if keep else None,
                                        s.data_ptr() if keep else None, o.data_ptr(), y2.data_ptr(), pre.data_ptr() if keep else None,
                                        mean.data_ptr(), rstd.data_ptr(), B, N, C, 0.25, 1e-5, torch.cuda.current_stream().cuda_stream), "half")
def three(keep):
    ee = dgf.row_gemm(yf, pe, C, C, bias=be)
    ss = torch.empty_like(ee)
    _lib.check(lib.dg_attn_core_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), ee.data_ptr(), ss.data_ptr(), o.data_ptr(), B, N, C, 0.25, 0,
                                    torch.cuda.current_stream().cuda_stream), "attn")
    return dgf.row_gemm(ss, po, C, C, bias=boe, residual=yf, ln=(g4, b4, 1e-5), want_pre=keep)
for keep in (True, False):
    tf, t3 = timeit(lambda: fused(keep)), timeit(lambda: three(keep))
    gb = 4 * R * C * (5 if keep else 2) / 1e9
    print(f"B = {B} keep = {keep}: fused {tf:7.1f} us ({gb / tf * 1e6 / 1e3:.2f} TB/s of {gb:.2f} GB)   three launches {t3:7.1f} us")
