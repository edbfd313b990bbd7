#!/usr/bin/env python
"""Phase times (s_memtime) of the K=384 row GEMM from a DG_DBG=16 build (developer tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import functional as dgf, _lib
K, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 128)
R = 256 * 45 * 45
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
print(f"wave time {us:.1f} us per launch -> {dbg.cpu().view(8, 8)[7, 4].item() / us / 1e3:.2f} GHz if ticks are shader cycles")
d = dbg.cpu().view(8, 8)
k384 = K == 384
names = {True: ["write(wait+split)", "fetch issue", "barrier", "store_tile"],
         False: ["mfma+frags", "fold", "exchange", "barrier"] if k384 else ["mfma+frags", "epilogue", "barrier", "-"]}
ncons = 4 if (k384 or N == 128) else 6
for w_ in range(8):
    r = d[w_].tolist()
    nm = names[w_ >= ncons]
    print(f"wave {w_} ({'mover' if w_ >= ncons else 'consumer'}): total {r[4]} ticks, tiles {r[5]}: " + ", ".join(f"{n} {v} ({100*v/max(r[4],1):.0f}%)" for n, v in zip(nm, r[:4])) + (f" | epilogue compute {r[6]} stores {r[7]}" if w_ < ncons and not k384 else ""))
