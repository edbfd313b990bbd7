#!/usr/bin/env python
"""Edge-level weight-gradient launches at configs[1] size (R = 518 400 rows, fp32): time per launch for the shapes
of the step (fp16 hi + lo planes with running column scales: the producer / consumer kernel of csrc/wgrad_stream.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from druggen_amd import functional as dgf

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 45 * 45
dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
torch.manual_seed(0)
for N, K in ((128, 128), (384, 128), (128, 384)):
    dy = (torch.randn(R, N, device="cuda") * 1e-3).to(dt)
    x = torch.randn(R, K, device="cuda").to(dt)
    for _ in range(3):
        dgf._wgrad(dy, x, True)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        dgf._wgrad(dy, x, True)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 20 * 1e3
    gb = (N + K) * dy.element_size() * R / 1e9
    print(f"wgrad {N}x{K} R={R}: {us:8.1f} us  {gb / us * 1e6:6.0f} GB/s")
