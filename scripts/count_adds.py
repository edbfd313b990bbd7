#!/usr/bin/env python
"""Which ATen elementwise adds / copies run inside one GAN step at configs[1] (torch profiler, grouped by shape)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from druggen_amd import synth
from druggen_amd.model import Discriminator, Generator
from druggen_amd.trainer import GANStep

dev = torch.device("cuda")
B, N, E, M = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 45, 5, 13
ctor = ("relu", N, E, M, 0.0)
kw = dict(dim=128, depth=4, heads=8, mlp_ratio=3)
torch.manual_seed(0)
G, D = Generator(*ctor, **kw).to(dev), Discriminator(*ctor, **kw).to(dev)
a, x, _, _ = synth.molecule_batch(B, N, E, M, seed=1)
da, dx, _, _ = synth.molecule_batch(B, N, E, M, seed=2)
ge, gn, de, dn = (torch.from_numpy(t).to(dev) for t in (a, x, da, dx))
stepper = GANStep(G, D, lambda_gp=10.0)
for _ in range(2):
    stepper.step(de, dn, ge, gn)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    stepper.step(de, dn, ge, gn)
torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::_foreach_add_", "aten::copy_", "aten::_foreach_copy_", "aten::mul", "aten::sum", "aten::zeros_like", "aten::zero_"):
        c[(e.name, str(e.input_shapes)[:70])] += 1
tot = collections.Counter()
for (n, sh), v in c.items():
    tot[n] += v
print(dict(tot))
for k, v in c.most_common(30):
    print(v, k)
