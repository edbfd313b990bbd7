#!/usr/bin/env python
"""Which weights take the single-launch pack path (dg_row_gemm_pack / _pack3) in a steady-state GAN step at BASELINE
configs[1] shapes (developer tool)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from druggen_amd import functional as dgf, synth, _lib
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
B = 32
a, x, _, _ = synth.molecule_batch(B, w["vertexes"], w["edges"], w["nodes"], seed=1234)
da, dx, _, _ = synth.molecule_batch(B, w["vertexes"], w["edges"], w["nodes"], seed=2234)
batch = [torch.from_numpy(t).to(dev) for t in (da, dx, a, x)]
stepper = GANStep(G, D, lambda_gp=10.0)
for _ in range(3):
    stepper.step(*batch)
names = {p.data_ptr(): n for m, pre in ((G, "G."), (D, "D.")) for n, p in m.named_parameters(prefix=pre)}
lib = _lib.load()
sites = collections.Counter()
for fn in ("dg_row_gemm_pack", "dg_row_gemm_pack3"):
    orig = getattr(lib, fn)
    def wrapped(*args, _orig=orig, _fn=fn):
        fr = [f for f in traceback.extract_stack() if os.sep + "functional" + os.sep in f.filename]
        who = " < ".join(f"{f.name}:{f.lineno}" for f in fr[-4:])
        ptr = args[0] if isinstance(args[0], int) else getattr(args[0], "value", args[0])
        sites[(_fn, names.get(ptr, hex(ptr) if isinstance(ptr, int) else str(ptr)), args[4] if _fn == "dg_row_gemm_pack" else args[5], who)] += 1
        return _orig(*args)
    setattr(lib, fn, wrapped)
stepper.step(*batch)
torch.cuda.synchronize()
print("single pack launches in one step:", sum(sites.values()))
for k, n in sites.most_common(40):
    print(n, k)
