import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druggen_amd import synth
from druggen_amd.model import Generator, Discriminator
from druggen_amd.trainer import GANStep
B, N, E, M, L = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 9, 5, 5, 1)))
torch.manual_seed(0)
G = Generator("relu", N, E, M, 0.0, dim=128, depth=L, heads=8, mlp_ratio=3).cuda()
D = Discriminator("relu", N, E, M, 0.0, dim=128, depth=L, heads=8, mlp_ratio=3).cuda()
a, x, _, _ = synth.molecule_batch(B, N, E, M, seed=1); da, dx, _, _ = synth.molecule_batch(B, N, E, M, seed=2)
a, x, da, dx = (torch.from_numpy(v).cuda() for v in (a, x, da, dx))
st = GANStep(G, D)
for _ in range(3): st.step(da, dx, a, x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): st.step(da, dx, a, x)
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
print(f"eager: {eager*1e3:.2f} ms/step  {B/eager:.0f} mol/s")
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): st.step(da, dx, a, x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = st.step(da, dx, a, x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 20
    print(f"graph: {gr*1e3:.2f} ms/step  {B/gr:.0f} mol/s  losses {out[0].item():.4f} {out[1].item():.4f}")
except Exception as ex:
    import traceback; traceback.print_exc()
