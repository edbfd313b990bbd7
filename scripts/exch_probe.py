import sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/scripts")
from druggen_amd import functional as dgf
from bench_kernels import timeit
R, K, N = 256 * 45 * 45, 128, 128
a = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
res = torch.randn(R, N, device="cuda"); g = torch.ones(N, device="cuda"); be = torch.zeros(N, device="cuda")
pw = dgf.packed_weight(w, 0)
print("direct            ", timeit(lambda: dgf.row_gemm(a, pw, K, N, bias=b)))
print("exch residual only", timeit(lambda: dgf.row_gemm(a, pw, K, N, bias=b, residual=res)))
print("exch residual+LN  ", timeit(lambda: dgf.row_gemm(a, pw, K, N, bias=b, residual=res, ln=(g, be, 1e-5))))
print("exch res+LN+pre   ", timeit(lambda: dgf.row_gemm(a, pw, K, N, bias=b, residual=res, ln=(g, be, 1e-5), want_pre=True)))
