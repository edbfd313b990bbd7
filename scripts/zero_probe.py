import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
from druggen_amd import functional as dgf
from bench_kernels import timeit
R = 256 * 45 * 45
for (K, N) in [(128, 128), (128, 384), (384, 128)]:
    for kind in ("randn", "zeros"):
        a = torch.randn(R, K, device="cuda") if kind == "randn" else torch.zeros(R, K, device="cuda")
        w = (torch.randn(N, K, device="cuda") * 0.05) if kind == "randn" else torch.zeros(N, K, device="cuda")
        pw = dgf.packed_weight(w, 0)
        t = timeit(lambda: dgf.row_gemm(a, pw, K, N))
        print(f"row_gemm K={K} N={N} {kind}: {t:.1f} us")
for (N, K) in [(128, 128), (384, 128)]:
    for kind in ("randn", "zeros"):
        dy = torch.randn(R, N, device="cuda") if kind == "randn" else torch.zeros(R, N, device="cuda")
        x = torch.randn(R, K, device="cuda") if kind == "randn" else torch.zeros(R, K, device="cuda")
        t = timeit(lambda: dgf._wgrad(dy, x, True))
        print(f"wgrad N={N} K={K} {kind}: {t:.1f} us")
