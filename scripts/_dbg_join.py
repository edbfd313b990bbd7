import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
import cases, harness
from test_hip_step import _build
from druggen_amd.model import gradient_penalty
name = sys.argv[1] if len(sys.argv) > 1 else "c2_b2"
case = cases.CASES[name]
inp = harness.torch_inputs(case, torch.float32, "cuda")
def grads(mode):
    os.environ["DG_PENALTY_WGRAD"] = mode
    cfg, G, D = _build(case)
    with torch.no_grad():
        _, _, ns, es = G(inp["gen_edge"], inp["gen_node"])
    gp = gradient_penalty(D, inp["disc_node"], inp["disc_edge"], ns, es, case["batch"], ns.device, eps=(inp["eps_edge"], inp["eps_node"]))
    params = [p for p in D.parameters()]
    names = [n for n, _ in D.named_parameters()]
    return names, [None if g is None else g.clone() for g in torch.autograd.grad(gp, params, allow_unused=True)]
for trial in range(2):
    n, a = grads("engine"); _, b = grads("joined"); _, c = grads("joined"); _, d = grads("engine")
    for nm, x, y, z, u in zip(n, a, b, c, d):
        if x is not None and not (torch.equal(x, y) and torch.equal(y, z) and torch.equal(x, u)):
            print(trial, nm, tuple(x.shape), "eng-join", float((x - y).abs().max() / x.abs().max()), "join-join", float((y - z).abs().max() / x.abs().max()), "eng-eng", float((x - u).abs().max() / x.abs().max()))
print("done")
