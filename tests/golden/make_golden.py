"""Generate golden vectors from the REFERENCE implementation.

Runs only in the build container (needs ``/root/reference``); the GPU box never
sees the reference.  For every case in ``cases.CASES`` it builds the reference's
own ``Generator`` / ``Discriminator`` (``src/model/models.py``), loads the
seeded weights, and executes one GAN iteration exactly as ``train.py:351-384``
does -- ``discriminator_loss`` -> backward -> AdamW, ``generator_loss`` ->
backward -> AdamW -- with the two ``torch.rand`` draws of
``src/model/loss.py:21-22`` replaced by the seeded eps.  Outputs, losses,
gradients and parameter updates are stored in float64 ("ref64", the numerical
truth) and float32 ("ref32", what the reference produces in practice).

    python tests/golden/make_golden.py            # rewrite all fixtures
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases  # noqa: E402

REFERENCE = os.environ.get("DRUGGEN_REFERENCE", "/root/reference")


def _import_reference():
    if not os.path.isdir(os.path.join(REFERENCE, "src", "model")):
        raise SystemExit(f"reference not found at {REFERENCE}")
    # the repo root also has a ``src`` package (the drop-in shim): make sure the
    # reference's one wins for this process.
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE)
    from src.model.models import Generator, Discriminator
    from src.model import loss as ref_loss
    sys.path.remove(REFERENCE)
    return Generator, Discriminator, ref_loss


class _RandQueue:
    """Stand-in for torch.rand that replays preset tensors in call order."""

    def __init__(self, tensors):
        self.q = list(tensors)

    def __call__(self, *shape, **kw):
        t = self.q.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.clone()


def run_reference(case: dict, dtype: torch.dtype):
    Generator, Discriminator, ref_loss = _import_reference()
    cfg = cases.net_config(case)
    torch.manual_seed(0)
    ctor = (cfg.act, cfg.vertexes, cfg.edges, cfg.nodes, cfg.dropout)
    kw = dict(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, mlp_ratio=cfg.mlp_ratio)
    G, D = Generator(*ctor, **kw), Discriminator(*ctor, **kw)
    gp_np, dp_np = cases.build_params(case)
    G.load_state_dict({k: torch.from_numpy(v) for k, v in gp_np.items()}, strict=True)
    D.load_state_dict({k: torch.from_numpy(v) for k, v in dp_np.items()}, strict=True)
    G, D = G.to(dtype), D.to(dtype)
    inp = {k: torch.from_numpy(v).to(dtype) for k, v in cases.build_inputs(case).items()}
    B, dev = case["batch"], torch.device("cpu")
    g_opt = torch.optim.AdamW(G.parameters(), 1e-5, [0.9, 0.999])   # train.py:213
    d_opt = torch.optim.AdamW(D.parameters(), 1e-5, [0.9, 0.999])   # train.py:214
    out = {}

    def snap(module):
        return {k: v.detach().clone() for k, v in module.state_dict().items()}

    # forward-only outputs
    with torch.no_grad():
        node, edge, node_sample, edge_sample = G(inp["gen_edge"], inp["gen_node"])
        out["G.node"], out["G.edge"] = node, edge
        out["G.node_sample"], out["G.edge_sample"] = node_sample, edge_sample
        out["D.real_logits"] = D(inp["disc_edge"], inp["disc_node"])
        out["D.fake_logits"] = D(edge_sample, node_sample)

    # gradient penalty on its own (loss.py:4-49)
    real_rand = torch.rand
    try:
        torch.rand = _RandQueue([inp["eps_edge"], inp["eps_node"]])
        out["gp"] = ref_loss.gradient_penalty(D, inp["disc_node"], inp["disc_edge"], node_sample, edge_sample,
                                              B, dev).detach()
        # ---- D step (train.py:352-368)
        g_opt.zero_grad(); d_opt.zero_grad()
        torch.rand = _RandQueue([inp["eps_edge"], inp["eps_node"]])
        _, _, d_loss = ref_loss.discriminator_loss(G, D, inp["disc_edge"], inp["disc_node"], inp["gen_edge"],
                                                   inp["gen_node"], B, dev, case["lambda_gp"])
    finally:
        torch.rand = real_rand
    d_loss.backward()
    out["d_loss"] = d_loss.detach()
    d_grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in D.named_parameters()}
    assert all(p.grad is None for p in G.parameters()), "reference G got grads in the D step"
    before = snap(D)
    d_opt.step()
    d_delta = {k: v - before[k] for k, v in snap(D).items()}
    # ---- G step (train.py:370-384)
    g_opt.zero_grad(); d_opt.zero_grad()
    g_loss = ref_loss.generator_loss(G, D, inp["gen_edge"], inp["gen_node"], B)[0]
    g_loss.backward()
    out["g_loss"] = g_loss.detach()
    g_grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in G.named_parameters()}
    before = snap(G)
    g_opt.step()
    g_delta = {k: v - before[k] for k, v in snap(G).items()}
    return out, d_grads, d_delta, g_grads, g_delta


def pack(case, tag, results, store):
    out, d_grads, d_delta, g_grads, g_delta = results
    for k, v in out.items():
        store[f"{tag}/{k}"] = v.double().numpy()
    store[f"{tag}/D.none_grads"] = np.array(json.dumps(sorted(k for k, v in d_grads.items() if v is None)))
    store[f"{tag}/G.none_grads"] = np.array(json.dumps(sorted(k for k, v in g_grads.items() if v is None)))
    for group, table in (("D.grad", d_grads), ("D.delta", d_delta), ("G.grad", g_grads), ("G.delta", g_delta)):
        for idx, (k, v) in enumerate(table.items()):
            if v is None:
                continue
            arr = v.double().numpy()
            store[f"{tag}/{group}/{k}"] = arr if case["full"] else cases.summarise(arr, idx)


def main(names=None):
    for name, case in cases.CASES.items():
        if names and name not in names:
            continue
        store = {"meta": np.array(json.dumps(dict(case=name, torch=torch.__version__,
                                                 reference="HUBioDataLab/DrugGEN snapshot 2025-09-26")))}
        pack(case, "ref64", run_reference(case, torch.float64), store)
        pack(case, "ref32", run_reference(case, torch.float32), store)
        if not case["full"]:
            for k in list(store):       # big activations -> summaries
                if k.endswith(("G.node", "G.edge")):
                    store[k] = cases.summarise(store[k], 999)
        np.savez_compressed(cases.fixture_path(name), **store)
        print(f"{name}: {os.path.getsize(cases.fixture_path(name)) / 1024:.0f} KiB  "
              f"d_loss={float(store['ref64/d_loss']):.6f} g_loss={float(store['ref64/g_loss']):.6f} "
              f"gp={float(store['ref64/gp']):.6f}")


if __name__ == "__main__":
    main(sys.argv[1:])
