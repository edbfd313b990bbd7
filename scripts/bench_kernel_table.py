#!/usr/bin/env python
"""Print the per-kernel table of a bench.py JSON line (stdin): launches per step, average us, share of the step."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f"{d['config'].get('baseline_config')} {d['dtype']}: {d['value']:.0f} mol/s  {d['ms_per_step']:.2f} ms")
for k, v in sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1].get("share_of_step", 0)):
    print(f"  {k:18s} {v['launches_per_step']:6.1f} x {v['avg_us']:8.1f} us = {v['launches_per_step'] * v['avg_us'] / 1e3:6.2f} ms  "
          f"{100 * v.get('share_of_step', 0):5.1f} %  {v.get('frac_of_hbm_peak', 0):.2f} of HBM")
