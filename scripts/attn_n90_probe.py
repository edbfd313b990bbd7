import os, sys, time, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/druggen_amd") else os.getcwd())
from druggen_amd import functional as dgf
B, N, C = 64, 90, 128
for dt in (torch.float32, torch.bfloat16):
    g = torch.Generator(device="cuda").manual_seed(0)
    mk = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.7).to(dt)
    q, k, v, e = mk(B, N, C), mk(B, N, C), mk(B, N, C), mk(B, N, N, C)
    ws, wo = mk(B, N, N, C), mk(B, N, C)
    tq, tk, tv, te = mk(B, N, C), mk(B, N, C), mk(B, N, C), mk(B, N, N, C)
    for name, fn in (("fwd", lambda: dgf._AttnCore.apply(q, k, v, e, 0.25, True)),
                     ("bwd", lambda: dgf._attn_bwd_launch(q, k, v, e, ws, wo, 0.25)),
                     ("bwd2", lambda: dgf._attn_bwd2_launch(q, k, v, e, ws, wo, tq, tk, tv, te, 0.25))):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print(f"N=90 {dt} attn_{name}: {(time.perf_counter() - t0) / 10 * 1e6:8.1f} us")
