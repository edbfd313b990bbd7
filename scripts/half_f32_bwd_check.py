#!/usr/bin/env python
"""dg_attn_half_f32_bwd1 (ln4 backward + ds = dz4 Woe + attention-core backward in one launch) against the two launches it
replaces (dg_row_gemm_ln_bwd_in + dg_attn_core_bwd) and their timing at BASELINE configs[1] shapes (developer tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import functional as dgf, _lib
from bench_kernels import timeit
lib = _lib.load()
torch.manual_seed(0)
C, alpha = 128, 0.25
for (B, N) in ((2, 9), (3, 45), (5, 48), (64, 45), (256, 45), (512, 45)):
    dev = "cuda"
    R = B * N * N
    dy2, pre = torch.randn(R, C, device=dev), torch.randn(R, C, device=dev) * 2 + 0.3
    mean = pre.mean(1).contiguous(); rstd = (1 / torch.sqrt(pre.var(1, unbiased=False) + 1e-5)).contiguous()
    g4 = torch.randn(C, device=dev) * 0.1 + 1
    woe = torch.randn(C, C, device=dev) * 0.1
    e = torch.randn(B, N, N, C, device=dev) * 0.5
    q, k, v, d_o = (torch.randn(B, N, C, device=dev) for _ in range(4))
    pwo = dgf.packed_weight(woe, 1)
    # the two launches
    dz_r, ds_r, dg_r, db_r = dgf.ln_bwd_row_gemm(pre, g4, mean, rstd, dy2, pwo, want_affine=True)
    dq_r, dk_r, dv_r, de_r = dgf._attn_bwd_launch(q, k, v, e, ds_r.view(B, N, N, C), d_o, alpha)
    # fused
    dz, de = torch.empty(R, C, device=dev), torch.empty(R, C, device=dev)
    dq, dk, dv = (torch.empty(B, N, C, device=dev) for _ in range(3))
    dgb = torch.empty(2, C, device=dev)
    ws = torch.empty(int(lib.dg_attn_half_f32_bwd1_workspace_bytes(B)), dtype=torch.uint8, device=dev)
    ds_out = torch.empty(R, C, device=dev)
    DS = None
    def fused():
        _lib.check(lib.dg_attn_half_f32_bwd1(dy2.data_ptr(), pre.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g4.data_ptr(), pwo.data_ptr(),
                                             e.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), d_o.data_ptr(), dz.data_ptr(), DS, de.data_ptr(),
                                             dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), ws.data_ptr(),
                                             ws.numel(), B, N, C, alpha, torch.cuda.current_stream().cuda_stream), "bwd1")
    fused(); torch.cuda.synchronize()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(B, N, "dz", rel(dz, dz_r), "de", rel(de.view_as(de_r), de_r), "dq", rel(dq, dq_r), "dk", rel(dk, dk_r), "dv", rel(dv, dv_r),
          "dgamma", rel(dgb[0], dg_r), "dbeta", rel(dgb[1], db_r))
    if B >= 256:
        def two():
            a_, b_, _, _ = dgf.ln_bwd_row_gemm(pre, g4, mean, rstd, dy2, pwo, want_affine=True)
            dgf._attn_bwd_launch(q, k, v, e, b_.view(B, N, N, C), d_o, alpha)
        t_plain = timeit(fused)
        DS = ds_out.data_ptr()
        print(f"   B = {B}: fused {t_plain:7.1f} us   fused + ds output {timeit(fused):7.1f} us   two launches {timeit(two):7.1f} us")
        DS = None
