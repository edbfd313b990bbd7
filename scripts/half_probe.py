"""Numerics + timing probe of the fused attention-half kernels against the unfused launches (developer tool)."""
import sys, time, math
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import _lib, functional as dgf

def ref_fwd(y, q, k, v, We, be, Woe, boe, g4, b4, alpha, eps, round_s):
    e = y @ We.t() + be
    s = alpha * q[:, :, None, :] * k[:, None, :, :] * (e * e + e)
    p = torch.softmax(s, dim=2)
    o = (p * v[:, None, :, :]).sum(2)
    s_in = s.to(round_s).to(s.dtype) if round_s is not None else s
    pre = y + s_in @ Woe.t() + boe
    y2 = torch.nn.functional.layer_norm(pre, (pre.shape[-1],), g4, b4, eps)
    return o, pre, y2

def main():
    B, N = int(sys.argv[1]), int(sys.argv[2])
    dtype = torch.bfloat16 if (len(sys.argv) < 4 or sys.argv[3] == "bf16") else torch.float32
    C, alpha, eps = 128, 0.25, 1e-5
    dev = "cuda"
    torch.manual_seed(0)
    y = (0.7 * torch.randn(B, N, N, C, device=dev)).to(dtype)
    q, k, v = (torch.randn(B, N, C, device=dev).to(dtype) for _ in range(3))
    We, Woe = (torch.randn(C, C, device=dev) / math.sqrt(C) for _ in range(2))
    be, boe, b4 = (0.1 * torch.randn(C, device=dev) for _ in range(3))
    g4 = 1 + 0.1 * torch.randn(C, device=dev)
    lib = _lib.load()
    code = _lib.DTYPES[dtype]
    packed = torch.empty(int(lib.dg_attn_half_packed_bytes(code)), dtype=torch.uint8, device=dev)
    _lib.check(lib.dg_attn_half_pack(We.data_ptr(), Woe.data_ptr(), packed.data_ptr(), code, None), "pack")
    o = torch.empty_like(q); y2 = torch.empty_like(y); pre = torch.empty_like(y)
    mean = torch.empty(B * N * N, device=dev); rstd = torch.empty_like(mean)
    stream = torch.cuda.current_stream().cuda_stream
    def run(edge=True):
        _lib.check(lib.dg_attn_half_fwd(y.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), packed.data_ptr(),
                                        be.data_ptr(), boe.data_ptr(), g4.data_ptr(), b4.data_ptr(), o.data_ptr(),
                                        y2.data_ptr() if edge else None, pre.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        B, N, C, alpha, eps, code, stream), "fwd")
    run()
    torch.cuda.synchronize()
    Wr = (lambda w_: w_.to(dtype).double()) if dtype == torch.bfloat16 else (lambda w_: w_.double())
    ro, rpre, ry2 = ref_fwd(y.double(), q.double(), k.double(), v.double(), Wr(We), be.double(), Wr(Woe), boe.double(),
                            g4.double(), b4.double(), alpha, eps, torch.bfloat16 if dtype == torch.bfloat16 else None)
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    print("o", rel(o, ro), "pre", rel(pre, rpre), "y2", rel(y2, ry2))
    mu = rpre.mean(-1).reshape(-1); rs = 1 / torch.sqrt(rpre.var(-1, unbiased=False) + eps).reshape(-1)
    print("mean", rel(mean, mu), "rstd", rel(rstd, rs))
    o.zero_(); run(False); torch.cuda.synchronize(); print("o (no edge)", rel(o, ro))
    for edge in (True, False):
        for _ in range(3): run(edge)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run(edge)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        nbytes = y.element_size() * B * N * N * C * (3 if edge else 1)
        print(f"fused fwd edge={edge}: {dt*1e6:.1f} us  {nbytes/dt/1e12:.2f} TB/s")
    # the unfused launches it replaces
    attn = torch.nn.Module()
    with dgf.activations(dtype):
        pw = lambda w_: dgf.packed_weight(w_, 0, dtype)
        yf = y.reshape(-1, C)
        def unfused():
            e = dgf.row_gemm(yf, pw(We), C, C, bias=be)
            s = torch.empty_like(e); oo = torch.empty_like(q)
            _lib.check(lib.dg_attn_core_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), e.data_ptr(), s.data_ptr(), oo.data_ptr(), B, N, C, alpha, code, stream), "a")
            return dgf.row_gemm(s, pw(Woe), C, C, bias=boe, residual=yf, ln=(g4, b4, eps), want_pre=True)
        for _ in range(3): unfused()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): unfused()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"unfused fwd: {dt*1e6:.1f} us")

main()
