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
with torch.no_grad(): _, _, ns, es = G(a, x)
for _ in range(4):
    for p in D.parameters(): p.grad = None
    (10 * gradient_penalty(D, dx, da, ns, es, B, "cuda")).backward()
torch.cuda.synchronize()
