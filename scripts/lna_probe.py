"""LayerNorm backward as the prologue of the 128 -> 128 input-gradient GEMM (dg_row_gemm_ln_bwd_in) against the two
launches it replaces: results and time.  usage: python scripts/lna_probe.py [R]"""
import sys
import torch
sys.path.insert(0, ".")
from druggen_amd import functional as F

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 45 * 45
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
pre = torch.randn(R, 128, device=dev, generator=g) * 1.7 + 0.3
dy = torch.randn(R, 128, device=dev, generator=g)
gamma = torch.rand(128, device=dev, generator=g) + 0.5
w = torch.randn(128, 128, device=dev, generator=g) * 0.1
mean = pre.mean(1)
rstd = (pre.var(1, unbiased=False) + 1e-5).rsqrt()
pk = F.packed_weight(w, 1, torch.float32)


def separate():
    dz, dg, db = F._ln_bwd_rows(pre, gamma, mean, rstd, dy)
    return dz, F.row_gemm(dz, pk, 128, 128), dg, db


def fused():
    return F.ln_bwd_row_gemm(pre, gamma, mean, rstd, dy, pk)


a, b = separate(), fused()
torch.cuda.synchronize()
for name, x, y in zip(("dz", "y", "dgamma", "dbeta"), a, b):
    print(name, "max |diff| / max |ref| =", float((x - y).abs().max() / x.abs().max()))
for name, fn in (("separate", separate), ("fused", fused), ("separate", separate), ("fused", fused)):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        fn()
    t1.record()
    torch.cuda.synchronize()
    print(f"{name:9s} R={R}: {t0.elapsed_time(t1) / 20 * 1e3:8.1f} us")
