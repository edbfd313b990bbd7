#!/usr/bin/env python
"""Host time per autograd node class in one GAN step at a small batch (developer tool): wall time spent inside the
forward / backward static methods of druggen_amd.functional's Functions (launch work, allocations, Python)."""
import collections, os, sys, time
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
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(cls, name):
    fn = getattr(cls, name)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[(cls.__name__, name)]
            e[0] += 1
            e[1] += time.perf_counter() - t0
    setattr(cls, name, staticmethod(timed))
for nm in dir(dgf):
    obj = getattr(dgf, nm)
    if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
        for m in ("forward", "backward"):
            if m in obj.__dict__:
                wrap(obj, m)
for _ in range(3):
    stepper.step(*batch)
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    stepper.step(*batch)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / n
print(f"B = {B}: {1e3 * tot:.1f} ms per step; inside Function methods {1e3 * sum(v[1] for v in acc.values()) / n:.1f} ms")
for (c, m), (k, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{1e3 * t / n:7.2f} ms  {k / n:6.1f} calls  {1e6 * t / k:7.1f} us each  {c}.{m}")
