#!/usr/bin/env python
"""Host-side cost of one GAN step (developer tool): cProfile over a few steps at a small batch, where the host's launch
work is the bound (BASELINE configs[1] shapes, B = 32)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from druggen_amd import functional as dgf, synth
from druggen_amd.model import Discriminator, Generator
from druggen_amd.trainer import GANStep
dev = torch.device("cuda", 0)
overrides, B, dtype, _ = bench.CONFIGS["c2"]
w = dict(bench.WORKLOAD, **overrides)
dgf.set_activation_dtype(dtype)
ctor = (w["act"], w["vertexes"], w["edges"], w["nodes"], w["dropout"])
kw = dict(dim=w["dim"], depth=w["depth"], heads=w["heads"], mlp_ratio=w["mlp_ratio"])
torch.manual_seed(0)
G, D = Generator(*ctor, **kw).to(dev), Discriminator(*ctor, **kw).to(dev)
B = int(os.environ.get("B", 32))
a, x, _, _ = synth.molecule_batch(B, w["vertexes"], w["edges"], w["nodes"], seed=1234)
da, dx, _, _ = synth.molecule_batch(B, w["vertexes"], w["edges"], w["nodes"], seed=2234)
batch = [torch.from_numpy(t).to(dev) for t in (da, dx, a, x)]
stepper = GANStep(G, D, lambda_gp=10.0)
for _ in range(3):
    stepper.step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    stepper.step(*batch)
t_host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 5
print(f"B = {B}: host returns after {1e3 * t_host:.1f} ms per step, GPU done after {1e3 * t_all:.1f} ms per step")
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):      # backward nodes on this thread: visible to cProfile
    pr.enable()
    for _ in range(3):
        stepper.step(*batch)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
