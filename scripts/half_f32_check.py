import os, sys, torch
sys.path.insert(0, '/root/repo')
from druggen_amd import functional as dgf, _lib
torch.manual_seed(0)
lib = _lib.load()
for (B, N) in ((2, 9), (3, 45), (5, 48), (64, 45), (256, 45)):
    C = 128
    dev = "cuda"
    y = torch.randn(B, N, N, C, device=dev)
    q, k, v = (torch.randn(B, N, C, device=dev) for _ in range(3))
    we, woe = torch.randn(C, C, device=dev) * 0.1, torch.randn(C, C, device=dev) * 0.1
    be, boe = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
    g4, b4 = torch.randn(C, device=dev) * 0.1 + 1, torch.randn(C, device=dev) * 0.1
    alpha, eps = 0.25, 1e-5
    R = B * N * N
    e, s, y2, pre = (torch.empty(R, C, device=dev) for _ in range(4))
    o = torch.empty(B, N, C, device=dev)
    mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
    for rep in range(3):
        st = lib.dg_attn_half_f32_fwd(y.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), dgf.packed_weight(we, 0).data_ptr(),
                                      be.data_ptr(), dgf.packed_weight(woe, 0).data_ptr(), boe.data_ptr(), g4.data_ptr(), b4.data_ptr(),
                                      e.data_ptr(), s.data_ptr(), o.data_ptr(), y2.data_ptr(), pre.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), B, N, C, alpha, eps, torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "half")
    torch.cuda.synchronize()
    yd = y.double().reshape(R, C)
    ed = yd @ we.double().t() + be.double()
    e4 = ed.view(B, N, N, C)
    sc = alpha * q.double()[:, :, None, :] * k.double()[:, None, :, :] * (e4 * e4 + e4)
    p = torch.softmax(sc, dim=2)
    od = (p * v.double()[:, None, :, :]).sum(2)
    pred = yd + sc.reshape(R, C) @ woe.double().t() + boe.double()
    mu = pred.mean(1, keepdim=True); var = pred.var(1, unbiased=False, keepdim=True)
    y2d = (pred - mu) / torch.sqrt(var + eps) * g4.double() + b4.double()
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    print(B, N, "e", rel(e, ed), "s", rel(s, sc.reshape(R, C)), "o", rel(o, od), "pre", rel(pre, pred), "y2", rel(y2, y2d),
          "mean", rel(mean, mu.squeeze(1)), "rstd", rel(rstd, 1 / torch.sqrt(var.squeeze(1) + eps)))
