#!/usr/bin/env python
"""Fused bf16 feed-forward (dg_ffn_ln_fwd_bf16 / _bwd_bf16) at configs[2] size: time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from druggen_amd import functional as dgf

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048 * 45 * 45
torch.manual_seed(0)
dev = "cuda"
x = torch.randn(R, 128, device=dev).bfloat16().requires_grad_(True)
w1 = (torch.randn(384, 128, device=dev) * 0.1).requires_grad_(True); b1 = (torch.randn(384, device=dev) * 0.1).requires_grad_(True)
w2 = (torch.randn(128, 384, device=dev) * 0.06).requires_grad_(True); b2 = (torch.randn(128, device=dev) * 0.1).requires_grad_(True)
g = (torch.rand(128, device=dev) + 0.5).requires_grad_(True); be = (torch.randn(128, device=dev) * 0.1).requires_grad_(True)
dy = torch.randn(R, 128, device=dev).bfloat16()
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(n): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n * 1e3
with torch.no_grad():
    t_f = timeit(lambda: dgf.ffn_ln(x, w1, b1, w2, b2, g, be, 1e-5))
y = dgf.ffn_ln(x, w1, b1, w2, b2, g, be, 1e-5)
t_dx = timeit(lambda: torch.autograd.grad(y, [x], dy, retain_graph=True))
t_all = timeit(lambda: torch.autograd.grad(y, [x, w1, b1, w2, b2, g, be], dy, retain_graph=True))
print(f"ffn bf16 R={R}: forward (no save) {t_f:7.1f} us   backward dx only {t_dx:7.1f} us   backward all {t_all:7.1f} us")
