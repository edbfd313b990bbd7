#!/usr/bin/env python
"""Golden case tiny_relu: gradients / AdamW deltas of the discriminator head against the fixture (developer tool)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden"))
import numpy as np, torch
import cases, harness
import test_hip_model as tm
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_relu"
case = cases.CASES[name]
fx = harness.load_fixture(name)
cfg, G, D = tm._build(case)
inp = harness.torch_inputs(case, torch.float32, "cuda")
res = harness.run_step(G, D, tm._d_loss, tm._g_loss, inp, case["lambda_gp"])
np.set_printoptions(linewidth=200, precision=4)
for k in res["D.grad"]:
    if not k.startswith("node_mlp"): continue
    for grp in ("D.grad", "D.delta"):
        got = res[grp][k]; want = fx[f"ref64/{grp}/{k}"]
        got = got.detach().cpu().numpy().astype(np.float64)
        e = np.abs(got - want).reshape(-1)
        print(grp, k, "rel", np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30), "worst idx", int(e.argmax()), "got", got.reshape(-1)[e.argmax()], "want", want.reshape(-1)[e.argmax()])
