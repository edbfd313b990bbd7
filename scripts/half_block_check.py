import sys, torch, math
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from druggen_amd import functional as dgf
from druggen_amd.model.layers import MHA
torch.manual_seed(3)
for need_edge in (True, False):
  for (B, N) in ((2, 7), (3, 45), (1, 20), (2, 90)):
    C, H = 128, 8
    attn = MHA(C, H).cuda()
    ln3, ln4 = torch.nn.LayerNorm(C).cuda(), torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        for ln in (ln3, ln4):
            ln.weight.add_(0.1 * torch.randn_like(ln.weight)); ln.bias.add_(0.1 * torch.randn_like(ln.bias))
    x1 = torch.randn(B, N, C, device="cuda").bfloat16().requires_grad_(True)
    y = (0.5 * torch.randn(B, N, N, C, device="cuda")).bfloat16().requires_grad_(True)
    params = [p for n_, p in attn.named_parameters() if need_edge or not n_.startswith("out_e")] + list(ln3.parameters()) + (list(ln4.parameters()) if need_edge else [])
    gouts = [torch.randn(B, N, C, device="cuda").bfloat16()] + ([torch.randn(B, N, N, C, device="cuda").bfloat16()] if need_edge else [])
    def run():
        x2, y2 = dgf.attn_block(x1, y, attn, ln3, ln4, need_edge)
        outs = (x2, y2) if need_edge else (x2,)
        g = torch.autograd.grad(outs, [x1, y] + params, gouts)
        return outs, g
    from druggen_amd.options import options
    with options.override(attn_half="fused"):
        of, gf = run()
    with options.override(attn_half="unfused"):
        ou, gu = run()
    # fp32 truth through the same module in float32 activations
    x1f, yf = x1.detach().float().requires_grad_(True), y.detach().float().requires_grad_(True)
    x2, y2 = dgf.attn_block(x1f, yf, attn, ln3, ln4, need_edge)
    outs = (x2, y2) if need_edge else (x2,)
    gt = torch.autograd.grad(outs, [x1f, yf] + params, [g.float() for g in gouts])
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
    worst_f = max(rel(a, b) for a, b in zip(list(of) + list(gf), list(outs) + list(gt)))
    worst_u = max(rel(a, b) for a, b in zip(list(ou) + list(gu), list(outs) + list(gt)))
    print(f"edge={need_edge} B={B} N={N}: fused vs fp32 {worst_f:.2e}   unfused-bf16 vs fp32 {worst_u:.2e}")
