import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_hip_kernels import _h16_chain
for R in (15, 33):
    L, dgf, t = _h16_chain(R)
    x, pw, C, H = t["x"], t["pw"], t["C"], t["H"]
    h32 = dgf.row_gemm(x, pw(t["w1"], 0), C, H, bias=t["b1"], relu=True)
    h16 = dgf.row_gemm(x, pw(t["w1"], 0), C, H, bias=t["b1"], relu=True, code=L.F32_H16)
    hd = dgf.hidden_to_float(h16, R)
    off = int(L.load().dg_hidden_scale_offset(R, 384))
    sc = h16[off:off + 4 * R].view(torch.float32)
    rowmax = h32.abs().amax(1)
    print("R", R, "scale*rowmax (want [2^-1? ...])", (rowmax / sc).tolist()[:16])
    err = (hd - h32).abs()
    bound = h32.abs() * 2.0 ** -11 + rowmax[:, None] * 2.0 ** -25
    ratio = err / bound
    i = int(ratio.argmax())
    r, c = i // H, i % H
    print("worst ratio", float(ratio.max()), "row", r, "col", c, float(hd[r, c]), float(h32[r, c]))
    print("rows failing", (ratio > 1).any(1).nonzero().flatten().tolist())
    print("cols failing in worst row", (ratio[r] > 1).nonzero().flatten().tolist()[:40])
