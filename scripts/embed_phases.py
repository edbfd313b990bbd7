#!/usr/bin/env python
"""Phase times (s_memtime) of the general edge-embedding backward from a -DDG_EMBED_DBG build (developer tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import functional as dgf, _lib
B, N, E = 256, 45, 5
a = torch.rand(B, N, N, E, device="cuda")
w1, b1 = torch.randn(64, E, device="cuda") * 0.3, torch.randn(64, device="cuda") * 0.1
w2, b2 = torch.randn(128, 64, device="cuda") * 0.1, torch.randn(128, device="cuda") * 0.1
g = torch.randn(B, N, N, 128, device="cuda")
lib = _lib.load()
need = int(lib.dg_embed_sym_workspace_bytes(B, N))
ws = torch.zeros(need // 4 + 16, device="cuda")
da = torch.empty_like(a)
dw1, db1, dw2, db2 = (torch.empty_like(t) for t in (w1, b1, w2, b2))
for use_da in (False, True):
    def run():
        _lib.check(lib.dg_embed_sym_bwd(a.data_ptr(), w1.data_ptr(), b1.data_ptr(), dgf._embed_packed_w2(w2).data_ptr(),
                                        dgf._embed_packed_w2(w2, True).data_ptr(), b2.data_ptr(), g.data_ptr(),
                                        da.data_ptr() if use_da else None, dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(),
                                        db2.data_ptr(), ws.data_ptr(), ws.numel() * 4, B, N, E, 64, 128, 0, 0,
                                        _lib.stream_of(a)), "bwd")
    for _ in range(3): run()
    torch.cuda.synchronize()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(10): run()
    e_.record(); torch.cuda.synchronize()
    us = s_.elapsed_time(e_) / 10 * 1e3
    tiles = B * ((N * (N + 1) // 2 + 31) // 32)
    grid = min(tiles, 512)
    d = ws.view(torch.int64)[((grid + 1) * 9408) // 2:][:32].view(4, 8).cpu()
    names = ["stage(a gather, layer 1)", "layer-2 MFMA", "g gather + dpre2", "dW2 MFMA", "dh MFMA", "dpre1 + dW1", "da + barrier"]
    for w_ in (0, 3):
        r = d[w_].tolist()
        print(f"da={use_da} {us:.0f} us  wave {w_}: total {r[7]} | " + ", ".join(f"{n} {100 * v / max(r[7], 1):.0f}%" for n, v in zip(names, r[:7])))
