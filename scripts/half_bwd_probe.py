"""Numerics + timing probe of dg_attn_half_bwd against autograd on the same math (developer tool)."""
import os, sys, time, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import _lib, functional as dgf


def main():
    B, N = int(sys.argv[1]), int(sys.argv[2])
    edge = (len(sys.argv) < 4 or sys.argv[3] != "noedge")
    dtype = torch.bfloat16
    C, alpha = 128, 0.25
    dev = "cuda"
    torch.manual_seed(0)
    y = (0.7 * torch.randn(B, N, N, C, device=dev)).to(dtype)
    q, k, v = (torch.randn(B, N, C, device=dev).to(dtype) for _ in range(3))
    dO = torch.randn(B, N, C, device=dev).to(dtype)
    dz = (torch.randn(B, N, N, C, device=dev)).to(dtype)
    We, Woe = (torch.randn(C, C, device=dev) / math.sqrt(C) for _ in range(2))
    be = 0.1 * torch.randn(C, device=dev)
    lib = _lib.load()
    code = _lib.DTYPES[dtype]
    packed = torch.empty(int(lib.dg_attn_half_packed_bytes(code)), dtype=torch.uint8, device=dev)
    _lib.check(lib.dg_attn_half_pack(We.data_ptr(), Woe.data_ptr(), packed.data_ptr(), code, None), "pack")
    dy = torch.empty_like(y); dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    dwe, dwoe = torch.zeros(C, C, device=dev), torch.zeros(C, C, device=dev)
    dbe, dboe = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.empty(int(lib.dg_attn_half_bwd_workspace_bytes(B, N)), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(wgrad=True):
        _lib.check(lib.dg_attn_half_bwd(y.data_ptr(), dz.data_ptr() if edge else None, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                        dO.data_ptr(), packed.data_ptr(), be.data_ptr(), dy.data_ptr(), dq.data_ptr(),
                                        dk.data_ptr(), dv.data_ptr(), dwe.data_ptr() if wgrad else None, dbe.data_ptr(),
                                        dwoe.data_ptr(), dboe.data_ptr(), ws.data_ptr(), ws.numel(), B, N, C, alpha, code,
                                        stream), "bwd")
    run()
    torch.cuda.synchronize()
    # reference: autograd in float64 on the GPU over bf16-rounded operands
    rd = torch.float64 if B * N * N <= 200000 else torch.float32
    f = lambda t: t.to(rd).requires_grad_(True)
    yr, qr, kr, vr = f(y), f(q), f(k), f(v)
    Wer, Woer, ber = f(We.to(dtype)), f(Woe.to(dtype)), f(be)
    boer = torch.zeros(C, device=dev, dtype=rd, requires_grad=True)
    e = yr @ Wer.t() + ber
    s = alpha * qr[:, :, None, :] * kr[:, None, :, :] * (e * e + e)
    p = torch.softmax(s, dim=2)
    o = (p * vr[:, None, :, :]).sum(2)
    loss = (o * dO.to(rd)).sum()
    if edge:
        pre = yr + s @ Woer.t() + boer
        loss = loss + (pre * dz.to(rd)).sum()
    loss.backward()
    rel = lambda a, b: float((a.to(rd) - b).norm() / b.norm())
    print("dy", rel(dy, yr.grad), "dq", rel(dq, qr.grad), "dk", rel(dk, kr.grad), "dv", rel(dv, vr.grad))
    print("dWe", rel(dwe, Wer.grad), "dbe", rel(dbe, ber.grad), end=" ")
    if edge:
        print("dWoe", rel(dwoe, Woer.grad), "dboe", rel(dboe, boer.grad))
    else:
        print()
    # reproducibility
    dy0, dk0, dwe0 = dy.clone(), dk.clone(), dwe.clone()
    run(); torch.cuda.synchronize()
    print("bit-reproducible:", torch.equal(dy0, dy) and torch.equal(dk0, dk) and torch.equal(dwe0, dwe))
    for wg in (True, False):
        for _ in range(3): run(wg)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run(wg)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        nbytes = 2 * B * N * N * C * (3 if edge else 2)
        print(f"fused bwd edge={edge} wgrad={wg}: {dt*1e6:.1f} us  {nbytes/dt/1e12:.2f} TB/s")

main()
