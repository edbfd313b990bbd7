import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("probe", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ffn_f32_probe.py"))
sys.argv = [sys.argv[0]]
probe = importlib.util.module_from_spec(spec); spec.loader.exec_module(probe)
from druggen_amd import _lib
_lib.load()
R = 518400
p = probe.params(5)
x = probe.gen((R, 128), 12).float().to("cuda").requires_grad_(True)
for keep in (True, False):
    xx = x if keep else x.detach()
    pp = p if keep else {k: v.detach() for k, v in p.items()}
    for _ in range(3):
        y, pre, mean, rstd = probe.run(xx, pp, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y, pre, mean, rstd = probe.run(xx, pp, True); e1.record(); torch.cuda.synchronize()
    t = mean[300000:300048].double().cpu().view(8, 6)
    print(os.environ.get("DG_LIB", "default").split("/")[-1], "keep" if keep else "nokeep", f"{e0.elapsed_time(e1) * 1000:.1f} us",
          "| ticks per wave (avg over 8 waves): fc1 wait/issue/compute, fc2 wait/issue/compute:", [int(v) for v in t.mean(0).tolist()],
          "| sum", int(t.mean(0).sum()))
