"""Numerics + timing of the bf16 edge-embedding backward (csrc/embed_bf16.hip) against float64 autograd and the general kernel."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import functional as dgf

B, N, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 5
act = sys.argv[4] if len(sys.argv) > 4 else "relu"
torch.manual_seed(0)
from druggen_amd.options import options      # noqa: E402
dev = "cuda"
a = torch.softmax(2 * torch.randn(B, N, N, E, device=dev), -1)
w1, b1 = torch.randn(64, E, device=dev) * 0.5, torch.randn(64, device=dev) * 0.1
w2, b2 = torch.randn(128, 64, device=dev) * 0.15, torch.randn(128, device=dev) * 0.1
g = torch.randn(B, N, N, 128, device=dev).bfloat16()

def run(mode, need_da=True):
    options.set(embed_bf16=mode)
    return dgf._embed_bwd_launch(a, w1, b1, w2, b2, g, act, torch.bfloat16, need_da, True)

fast = run("fast"); torch.cuda.synchronize()
gen = run("general"); torch.cuda.synchronize()
rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
if B * N * N <= 300000:
    ad = a.double().requires_grad_(True)
    ws = [t.double().requires_grad_(True) for t in (w1, b1, w2, b2)]
    f = torch.relu if act == "relu" else (lambda t: torch.nn.functional.leaky_relu(t, 0.01))
    ee = f(torch.nn.functional.linear(f(torch.nn.functional.linear(ad, ws[0], ws[1])), ws[2], ws[3]))
    out = (ee + ee.permute(0, 2, 1, 3)) / 2
    ref = torch.autograd.grad(out, [ad] + ws, g.double())
    for name, x, y, z in zip("da dw1 db1 dw2 db2".split(), fast, gen, ref):
        print(f"{name}: fast vs fp64 {rel(x, z):.2e}   general vs fp64 {rel(y, z):.2e}")
else:
    for name, x, y in zip("da dw1 db1 dw2 db2".split(), fast, gen):
        print(f"{name}: fast vs general {rel(x, y):.2e}")
for mode in ("fast", "general"):
    for da in (True, False):
        for _ in range(2): run(mode, da)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): run(mode, da)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{mode} need_da={da}: {dt*1e6:.0f} us   {2 * B*N*N*128 / dt / 1e12:.2f} TB/s of one pass over g")
