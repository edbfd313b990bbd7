import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import synth
from druggen_amd.model import Generator, Discriminator, gradient_penalty
B, N, E, M, L = 256, 45, 5, 13, 4
torch.manual_seed(0)
G = Generator("relu", N, E, M, 0.0, dim=128, depth=L, heads=8, mlp_ratio=3).cuda()
D = Discriminator("relu", N, E, M, 0.0, dim=128, depth=L, heads=8, mlp_ratio=3).cuda()
a, x, _, _ = synth.molecule_batch(B, N, E, M, seed=1); da, dx, _, _ = synth.molecule_batch(B, N, E, M, seed=2)
a, x, da, dx = (torch.from_numpy(v).cuda() for v in (a, x, da, dx))
def T(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
def zero():
    for p in list(G.parameters()) + list(D.parameters()): p.grad = None
print("D fwd (no grad):        %.2f ms" % T(lambda: torch.no_grad().__enter__() or D(da, dx)))
torch.set_grad_enabled(True)
print("G fwd (no grad):        %.2f ms" % T(lambda: [torch.set_grad_enabled(False), G(a, x), torch.set_grad_enabled(True)]))
def d_real():
    zero(); (-D(da, dx).mean()).backward()
print("D fwd+bwd (real):       %.2f ms" % T(d_real))
with torch.no_grad(): _, _, ns, es = G(a, x)
def gp():
    zero(); (10 * gradient_penalty(D, dx, da, ns, es, B, "cuda")).backward()
print("GP fwd+grad+dbl bwd:    %.2f ms" % T(gp))
def gstep():
    zero()
    for p in D.parameters(): p.requires_grad_(False)
    _, _, n2, e2 = G(a, x); (-D(e2, n2).mean()).backward()
    for p in D.parameters(): p.requires_grad_(True)
print("G step (G+D fwd, bwd):  %.2f ms" % T(gstep))
