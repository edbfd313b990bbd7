"""Stress check of the producer / consumer row GEMMs against fp64 (developer tool): repeated launches, several row counts."""
import sys, torch
sys.path.insert(0, '/root/repo')
from druggen_amd import functional as dgf
torch.manual_seed(0)
for (K, N) in ((384, 128), (128, 384)):
    tot_bad = 0
    for R in (1, 16, 17, 63, 64, 200, 4097, 518400):
        a = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
        pw = dgf.packed_weight(w, 0)
        ref = (a.double() @ w.double().t() + b.double())
        reps = 3 if R > 100000 else 12
        for it in range(reps):
            relu = bool(it & 1)
            y = dgf.row_gemm(a, pw, K, N, bias=b, relu=relu)
            want = torch.relu(ref) if relu else ref
            err = (y.double() - want).abs()
            bad = (err > 1e-4 * (1 + want.abs())).nonzero()
            tot_bad += bad.shape[0]
            if bad.shape[0]:
                print("BAD", K, N, R, relu, float(err.max()), bad.shape[0], bad[:4].tolist())
    print(f"K={K} N={N}: {tot_bad} bad elements")
