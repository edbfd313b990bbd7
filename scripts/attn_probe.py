#!/usr/bin/env python
"""Run the three attention-core kernels a few times at a given batch / dtype (for rocprofv3 --pmc passes and timing).
    python scripts/attn_probe.py [B=256] [dtype=f32|bf16] [reps=5]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import functional as dgf   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, C = 45, 128
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.7).to(dt)
q, k, v, e = mk(B, N, C), mk(B, N, C), mk(B, N, C), mk(B, N, N, C)
ws, wo = mk(B, N, N, C), mk(B, N, C)
tq, tk, tv, te = mk(B, N, C), mk(B, N, C), mk(B, N, C), mk(B, N, N, C)
for name, fn in (("fwd", lambda: dgf._AttnCore.apply(q, k, v, e, 0.25, True)),
                 ("bwd", lambda: dgf._attn_bwd_launch(q, k, v, e, ws, wo, 0.25)),
                 ("bwd+add_e", lambda: dgf._attn_bwd_launch(q, k, v, e, ws, wo, 0.25, add_e=te)),
                 ("bwd2", lambda: dgf._attn_bwd2_launch(q, k, v, e, ws, wo, tq, tk, tv, te, 0.25))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    es = q.element_size()
    nb = {"fwd": es * B * (2 * N * N * C + 4 * N * C), "bwd": es * B * (3 * N * N * C + 7 * N * C),
          "bwd+add_e": es * B * (4 * N * N * C + 7 * N * C), "bwd2": es * B * (5 * N * N * C + 11 * N * C)}[name]
    print(f"attn_{name} B={B} {dt}: {us:9.1f} us  {nb / us / 1e6:7.2f} TB/s algorithmic")
