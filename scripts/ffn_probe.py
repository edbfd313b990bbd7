#!/usr/bin/env python
"""Time the bf16 feed-forward kernels (fused vs unfused) at a given number of rows; also the target of PMC passes.
    python scripts/ffn_probe.py [B=2048] [reps=5]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import _lib, functional as dgf   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
R = B * 45 * 45
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.randn(R, 128, device="cuda", generator=g)).to(torch.bfloat16).requires_grad_(True)
dy = (torch.randn(R, 128, device="cuda", generator=g)).to(torch.bfloat16)
ps = [(torch.randn(384, 128, device="cuda", generator=g) * 0.1).requires_grad_(True), torch.zeros(384, device="cuda").requires_grad_(True),
      (torch.randn(128, 384, device="cuda", generator=g) * 0.06).requires_grad_(True), torch.zeros(128, device="cuda").requires_grad_(True),
      torch.ones(128, device="cuda").requires_grad_(True), torch.zeros(128, device="cuda").requires_grad_(True)]


def timed(fn, what, nbytes):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"{what:44s} {us:9.1f} us   {nbytes / us / 1e6:6.2f} TB/s algorithmic")


for mode in ("fused", "unfused"):
    os.environ["DG_FFN_BF16"] = mode
    y = dgf.ffn_ln(x, *ps, 1e-5)
    timed(lambda: dgf.ffn_ln(x, *ps, 1e-5), f"{mode} forward (saves for backward)", 2 * R * 128 * 3)
    with torch.no_grad():
        timed(lambda: dgf.ffn_ln(x, *ps, 1e-5), f"{mode} forward (no_grad)", 2 * R * 128 * 2)
    timed(lambda: torch.autograd.grad(y, [x], dy, retain_graph=True), f"{mode} backward, dx only", 2 * R * 128 * 4)
    timed(lambda: torch.autograd.grad(y, [x] + ps, dy, retain_graph=True), f"{mode} backward, dx + weights", 2 * R * 128 * 8)
_lib.prof_reset()
