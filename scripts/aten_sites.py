#!/usr/bin/env python
"""Which call sites launch the ATen / copy kernels of a GAN step (developer tool): torch.profiler with stacks over ONE
step at the bench workload (BASELINE configs[1]); leaf ATen ops that ran a device kernel, grouped by op, innermost
druggen_amd frame and enclosing autograd node."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

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
B = int(os.environ.get("B", B))
a, x, _, _ = synth.molecule_batch(B, w["vertexes"], w["edges"], w["nodes"], seed=1234)
da, dx, _, _ = synth.molecule_batch(B, w["vertexes"], w["edges"], w["nodes"], seed=2234)
batch = [torch.from_numpy(t).to(dev) for t in (da, dx, a, x)]
stepper = GANStep(G, D, lambda_gp=10.0)
for _ in range(2):
    stepper.step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    stepper.step(*batch)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
        continue
    if any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue      # not a leaf
    site = ""
    for fr in ev.stack or []:
        if "druggen_amd" in fr or "bench.py" in fr:
            site = fr.split("/")[-1]
            break
    node, p = "", ev.cpu_parent
    while p is not None:
        if "Backward" in p.name or "evaluate_function" in p.name or "AccumulateGrad" in p.name:
            node = p.name.replace("autograd::engine::evaluate_function: ", "")
            break
        p = p.cpu_parent
    agg[(ev.name, site or str(getattr(ev, 'input_shapes', ''))[:60], node)] += 1
print(f"# leaf ATen ops with device time in one step (B = {B}): {sum(agg.values())}")
for (name, site, node), n in agg.most_common(70):
    print(f"{n:5d}  {name:30s} {site:50s} {node}")
# device-to-device copies (hipMemcpyAsync: clone / contiguous / copy_ of contiguous tensors), by enclosing op chain
cp = collections.Counter()
for ev in prof.events():
    if "emcpy" not in ev.name or ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    chain, p = [], ev.cpu_parent
    while p is not None and len(chain) < 4:
        chain.append(p.name.replace("autograd::engine::evaluate_function: ", ""))
        p = p.cpu_parent
    cp[(ev.name, " < ".join(chain))] += 1
print(f"# memcpy runtime calls in the step: {sum(cp.values())}")
for (name, chain), n in cp.most_common(30):
    print(f"{n:5d}  {name:24s} {chain}")
