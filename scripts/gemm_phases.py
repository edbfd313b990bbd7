#!/usr/bin/env python
"""Phase times (s_memtime) of the K=384 row GEMM from a DG_DBG=16 build (developer tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import functional as dgf, _lib
R, K, N = 256 * 45 * 45, 384, 128
a = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
pw = dgf.packed_weight(w, 0)
y = torch.empty(R, N, device="cuda")
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
lib = _lib.load()
import time
def run():
    _lib.check(lib.dg_row_gemm(a.data_ptr(), pw.data_ptr(), y.data_ptr(), R, K, N, b.data_ptr(), 0, None, None, None, None, None,
                               None, dbg.data_ptr(), None, 0.0, 0, _lib.stream_of(a)), "dg_row_gemm")
for _ in range(3): run()
torch.cuda.synchronize()
s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s_.record()
for _ in range(20): run()
e_.record(); torch.cuda.synchronize()
us = s_.elapsed_time(e_) / 20 * 1e3
print(f"wave time {us:.1f} us per launch -> {dbg.cpu().view(8, 8)[4, 4].item() / us / 1e3:.2f} GHz if ticks are shader cycles")
d = dbg.cpu().view(8, 8)
names = {True: ["write(wait+split)", "fetch issue", "barrier", "store_tile"], False: ["mfma+frags", "fold", "exchange", "barrier"]}
for w_ in range(8):
    r = d[w_].tolist()
    nm = names[w_ >= 4]
    print(f"wave {w_} ({'mover' if w_ >= 4 else 'consumer'}): total {r[4]} ticks, tiles {r[5]}: " + ", ".join(f"{n} {v} ({100*v/max(r[4],1):.0f}%)" for n, v in zip(nm, r[:4])))
