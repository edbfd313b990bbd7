"""DG_DTYPE_F32_H16 kernels (fp16 hidden plane + row scales) against fp64 and against the float32 kernels, with timings.
    python scripts/h16_probe.py [R]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from druggen_amd import _lib, functional as dgf      # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 518400
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(1)
C, H = 128, 384
x = torch.randn(R, C, device=dev, generator=g)
w1 = torch.randn(H, C, device=dev, generator=g) * 0.09
b1 = torch.randn(H, device=dev, generator=g) * 0.1
w2 = torch.randn(C, H, device=dev, generator=g) * 0.05
b2 = torch.randn(C, device=dev, generator=g) * 0.1
gam = torch.rand(C, device=dev, generator=g) + 0.5
bet = torch.randn(C, device=dev, generator=g) * 0.1
dz = torch.randn(R, C, device=dev, generator=g) * 1e-4
pw = lambda w, m: dgf.packed_weight(w, m, torch.float32)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


res = {}
for name, code in (("f32", 0), ("h32", _lib.F32_H32), ("h24", _lib.F32_H24), ("h16", _lib.F32_H16)):
    # fc1: h = relu(x W1^T + b1)
    h, bits = dgf.row_gemm(x, pw(w1, 0), C, H, bias=b1, relu=True, want_relu_bits=True, code=code)
    hf = dgf.hidden_to_float(h, R) if code else h
    # fc2 + residual + LN
    y, mean, rstd, pre = dgf.row_gemm(h, pw(w2, 0), H, C, bias=b2, residual=x, ln=(gam, bet, 1e-5), want_pre=True, R=R)
    # dh = (dz W2) * mask ; dx = dz + dh W1
    dh = dgf.row_gemm(dz, pw(w2, 1), C, H, mask_bits=bits, code=code)
    dhf = dgf.hidden_to_float(dh, R) if code else dh
    dx = dgf.row_gemm(dh, pw(w1, 1), H, C, residual=dz, R=R)
    dw2, db2 = dgf._wgrad(dz, h, True)
    dw1, db1 = dgf._wgrad(dh, x, True)
    res[name] = dict(h=hf, pre=pre, y=y, dh=dhf, dx=dx, dw2=dw2, db2=db2, dw1=dw1, db1=db1, bits=bits)
    t = {}
    t["n384 fwd"] = timeit(lambda: dgf.row_gemm(x, pw(w1, 0), C, H, bias=b1, relu=True, want_relu_bits=True, code=code))
    t["k384 +res+LN+pre"] = timeit(lambda: dgf.row_gemm(h, pw(w2, 0), H, C, bias=b2, residual=x, ln=(gam, bet, 1e-5), want_pre=True, R=R))
    t["n384 mask-in"] = timeit(lambda: dgf.row_gemm(dz, pw(w2, 1), C, H, mask_bits=bits, code=code))
    t["k384 +res"] = timeit(lambda: dgf.row_gemm(dh, pw(w1, 1), H, C, residual=dz, R=R))
    t["wgrad 128x384 (dz^T h)"] = timeit(lambda: dgf._wgrad(dz, h, True))
    t["wgrad 384x128 (dh^T x)"] = timeit(lambda: dgf._wgrad(dh, x, True))
    res[name]["t"] = t
    print(name, {k: round(v, 1) for k, v in t.items()}, flush=True)

# fp64 truth on a slice of rows (all rows for the weight gradients: chunks)
n = min(R, 20000)
xd, dzd = x[:n].double(), dz[:n].double()
hd = torch.relu(xd @ w1.double().t() + b1.double())
pred = xd + hd @ w2.double().t() + b2.double()
yd = torch.nn.functional.layer_norm(pred, (C,), gam.double(), bet.double(), 1e-5)
dhd = (dzd @ w2.double()) * (hd > 0)
dxd = dzd + dhd @ w1.double()
nw = (R + 31) // 32 * 512      # words the kernel writes (the allocation is an upper bound over kernel variants)
assert torch.equal(res["f32"]["bits"][:nw], res["h16"]["bits"][:nw]), "ReLU masks differ"
for name in ("f32", "h32", "h24", "h16"):
    r = res[name]
    print(name, "h", f"{rel(r['h'][:n], hd):.2e}", "pre", f"{rel(r['pre'][:n], pred):.2e}", "y", f"{rel(r['y'][:n], yd):.2e}",
          "dh", f"{rel(r['dh'][:n], dhd):.2e}", "dx", f"{rel(r['dx'][:n], dxd):.2e}")
# weight gradients against fp64 over all rows, chunked
dw2d = torch.zeros(C, H, dtype=torch.float64, device=dev)
dw1d = torch.zeros(H, C, dtype=torch.float64, device=dev)
db1d = torch.zeros(H, dtype=torch.float64, device=dev)
for i in range(0, R, 65536):
    xs, dzs = x[i:i + 65536].double(), dz[i:i + 65536].double()
    hs = torch.relu(xs @ w1.double().t() + b1.double())
    dhs = (dzs @ w2.double()) * (hs > 0)
    dw2d += dzs.t() @ hs
    dw1d += dhs.t() @ xs
    db1d += dhs.sum(0)
for name in ("f32", "h32", "h24", "h16"):
    r = res[name]
    print(name, "dW2", f"{rel(r['dw2'], dw2d):.2e}", "db2", f"{rel(r['db2'], dz.double().sum(0)):.2e}", "dW1", f"{rel(r['dw1'], dw1d):.2e}",
          "db1", f"{rel(r['db1'], db1d):.2e}")
# repeatability + odd row counts
for Rr in (1, 15, 16, 17, 33, 1000, 4097):
    xs, dzs = x[:Rr].contiguous(), dz[:Rr].contiguous()
    h, bits = dgf.row_gemm(xs, pw(w1, 0), C, H, bias=b1, relu=True, want_relu_bits=True, code=_lib.F32_H16)
    y = dgf.row_gemm(h, pw(w2, 0), H, C, bias=b2, residual=xs, R=Rr)
    hd = torch.relu(xs.double() @ w1.double().t() + b1.double())
    yd = xs.double() + hd @ w2.double().t() + b2.double()
    dw2, db2 = dgf._wgrad(dzs, h, True)
    dh = dgf.row_gemm(dzs, pw(w2, 1), C, H, mask_bits=bits, code=_lib.F32_H16)
    dw1, db1 = dgf._wgrad(dh, xs, True)
    dhd = (dzs.double() @ w2.double()) * (hd > 0)
    print("R", Rr, "h", f"{rel(dgf.hidden_to_float(h, Rr), hd):.2e}", "y", f"{rel(y, yd):.2e}", "dW2", f"{rel(dw2, dzs.double().t() @ hd):.2e}",
          "dW1", f"{rel(dw1, dhd.t() @ xs.double()):.2e}", "db1", f"{rel(db1, dhd.sum(0)):.2e}")
