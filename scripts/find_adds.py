#!/usr/bin/env python
"""Which autograd nodes cause the large ATen elementwise adds of a bf16 step (gradient accumulation of a tensor with two
consumers)?  One step under torch.profiler; every aten::add / add_ over >= 1e6 elements is printed with its enclosing events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from druggen_amd import functional as dgf
from druggen_amd.model import Generator, Discriminator
from druggen_amd.trainer import GANStep
from druggen_amd import synth

dev = torch.device("cuda:0")
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dgf.set_activation_dtype(dt)
w = dict(bench.WORKLOAD)
ctor = (w["act"], w["vertexes"], w["edges"], w["nodes"], w["dropout"])
kw = dict(dim=w["dim"], depth=w["depth"], heads=w["heads"], mlp_ratio=w["mlp_ratio"])
torch.manual_seed(0)
G, D = Generator(*ctor, **kw).to(dev), Discriminator(*ctor, **kw).to(dev)
a, x, _, _ = synth.molecule_batch(B, w["vertexes"], w["edges"], w["nodes"], seed=1)
da, dx, _, _ = synth.molecule_batch(B, w["vertexes"], w["edges"], w["nodes"], seed=2)
ge, gn, de, dn = (torch.from_numpy(t).to(dev) for t in (a, x, da, dx))
st = GANStep(G, D, lambda_gp=10.0)
for _ in range(2):
    st.step(de, dn, ge, gn)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    st.step(de, dn, ge, gn)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ("aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::copy_", "aten::to", "aten::_to_copy", "aten::cat", "aten::clone", "aten::contiguous"):
        shapes = ev.input_shapes
        n = 1
        for d in (shapes[0] if shapes and shapes[0] else []):
            n *= d
        if n >= 1_000_000:
            chain = []
            p = ev.cpu_parent
            while p is not None and len(chain) < 4:
                chain.append(p.name)
                p = p.cpu_parent
            print(ev.name, shapes[:2], "<-", " <- ".join(chain))
