"""Time of the fused float32 feed-forward forward through its autograd node at R = 518 400, with and without the tensors a backward
keeps: one line per library build (DG_LIB=<ablation build of scripts/build_variant.sh>; scripts/ffn_f32_variants.sh runs them all)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("probe", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ffn_f32_probe.py"))
probe = importlib.util.module_from_spec(spec); spec.loader.exec_module(probe)
from druggen_amd import _lib
_lib.load()
R = 518400
p = probe.params(5)
x = probe.gen((R, 128), 12).float().to("cuda").requires_grad_(True)
out = []
for keep in (True, False):
    xx = x if keep else x.detach()
    pp = p if keep else {k: v.detach() for k, v in p.items()}
    for _ in range(10):
        probe.run(xx, pp, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        probe.run(xx, pp, True)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) * 1000 / 40)
print(f"{os.environ.get('DG_LIB', 'default').split('/')[-1]:14s} keep {out[0]:7.1f} us   nokeep {out[1]:7.1f} us", flush=True)
