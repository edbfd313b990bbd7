"""Shared machinery of the parity tests: run one GAN iteration with any
(G, D, loss functions) triple and compare against a golden fixture."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import cases  # noqa: E402


def load_fixture(name):
    z = np.load(cases.fixture_path(name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def run_step(G, D, d_loss_fn, g_loss_fn, inputs, lambda_gp, lr=1e-5):
    """train.py:351-384 for modules G, D.

    d_loss_fn(G, D, disc_edge, disc_node, gen_edge, gen_node, lambda_gp, eps_edge, eps_node) -> d_loss
    g_loss_fn(G, D, gen_edge, gen_node) -> g_loss
    Returns dict with losses, grads ({name: tensor|None}) and AdamW deltas.
    """
    g_opt = torch.optim.AdamW(G.parameters(), lr, (0.9, 0.999))
    d_opt = torch.optim.AdamW(D.parameters(), lr, (0.9, 0.999))
    res = {}
    g_opt.zero_grad(set_to_none=True); d_opt.zero_grad(set_to_none=True)
    d_loss = d_loss_fn(G, D, inputs["disc_edge"], inputs["disc_node"], inputs["gen_edge"], inputs["gen_node"],
                       lambda_gp, inputs["eps_edge"], inputs["eps_node"])
    d_loss.backward()
    res["d_loss"] = d_loss.detach()
    res["D.grad"] = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in D.named_parameters()}
    res["G.grad_in_d_step"] = [k for k, p in G.named_parameters() if p.grad is not None]
    before = {k: p.detach().clone() for k, p in D.named_parameters()}
    d_opt.step()
    res["D.delta"] = {k: p.detach() - before[k] for k, p in D.named_parameters()}
    g_opt.zero_grad(set_to_none=True); d_opt.zero_grad(set_to_none=True)
    g_loss = g_loss_fn(G, D, inputs["gen_edge"], inputs["gen_node"])
    g_loss.backward()
    res["g_loss"] = g_loss.detach()
    res["G.grad"] = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in G.named_parameters()}
    before = {k: p.detach().clone() for k, p in G.named_parameters()}
    g_opt.step()
    res["G.delta"] = {k: p.detach() - before[k] for k, p in G.named_parameters()}
    return res


def _np(t):
    return t.detach().double().cpu().numpy()


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = np.linalg.norm(b.reshape(-1))
    return float(np.linalg.norm((a - b).reshape(-1)) / (den if den > 0 else 1.0))


def compare_scalar(got, want, rtol, what):
    got = float(got.detach()) if torch.is_tensor(got) else float(got)
    want = float(want)
    assert abs(got - want) <= rtol * max(1.0, abs(want)), f"{what}: got {got!r} want {want!r}"


def grad_table_errors(case, fixture, tag, group, table):
    """Per-tensor relative errors of {name: tensor|None} against the fixture group (full tensors or
    summaries), sorted worst first.  Error model (hard part 4): per tensor
    ||got - want|| / max(||want||, ||all grads|| / sqrt(n_tensors))."""
    want_none = set(json.loads(str(fixture[f"{tag}/{group.split('.')[0]}.none_grads"]))) if group.endswith("grad") else set()
    names = list(table.keys())
    got_none = {k for k, v in table.items() if v is None}
    if group.endswith("grad"):
        assert got_none == want_none, f"{group}: None-grad sets differ: {sorted(got_none ^ want_none)}"
    full = case["full"]
    wants, gots = {}, {}
    for idx, k in enumerate(names):
        if table[k] is None:
            continue
        w = fixture[f"{tag}/{group}/{k}"]
        g = _np(table[k])
        gots[k] = g if full else cases.summarise(g, idx)
        wants[k] = w
    if full:
        total = np.sqrt(sum(float((w ** 2).sum()) for w in wants.values()))
    else:
        total = np.sqrt(sum(float(w[0] ** 2) for w in wants.values()))
    floor = total / np.sqrt(max(1, len(wants)))
    errs = []
    for k in wants:
        if full:
            err = np.linalg.norm((gots[k] - wants[k]).reshape(-1))
            scale = max(np.linalg.norm(wants[k].reshape(-1)), floor)
        else:
            # summary = [norm, 4 projections, first, last]; projections of a
            # tensor with n elements have magnitude ~ norm * sqrt(n)/sqrt(3)/sqrt(n) ~ norm
            err = np.abs(gots[k] - wants[k]).max()
            scale = max(wants[k][0], floor)
        errs.append((float(err / scale if scale > 0 else err), k))
    errs.sort(reverse=True)
    return errs


def compare_grad_table(case, fixture, tag, group, table, rtol, name_map=None):
    """Assert every tensor of the table within ``rtol`` (see grad_table_errors); returns the worst."""
    errs = grad_table_errors(case, fixture, tag, group, table)
    for r, k in errs:
        assert r <= rtol, f"{group}/{k}: rel err {r:.3e} > {rtol:.1e}"
    return errs[0] if errs else (0.0, None)


def compare_step(case, fixture, tag, res, rtol_loss, rtol_grad, rtol_delta=None):
    compare_scalar(res["d_loss"], fixture[f"{tag}/d_loss"], rtol_loss, "d_loss")
    compare_scalar(res["g_loss"], fixture[f"{tag}/g_loss"], rtol_loss, "g_loss")
    assert res["G.grad_in_d_step"] == [], "generator received gradients in the D step"
    worst = {}
    worst["D.grad"] = compare_grad_table(case, fixture, tag, "D.grad", res["D.grad"], rtol_grad)
    worst["G.grad"] = compare_grad_table(case, fixture, tag, "G.grad", res["G.grad"], rtol_grad)
    if rtol_delta is not None:
        worst["D.delta"] = compare_grad_table(case, fixture, tag, "D.delta", res["D.delta"], rtol_delta)
        worst["G.delta"] = compare_grad_table(case, fixture, tag, "G.delta", res["G.delta"], rtol_delta)
    return worst


def torch_inputs(case, dtype, device="cpu"):
    return {k: torch.from_numpy(v).to(device=device, dtype=dtype) for k, v in cases.build_inputs(case).items()}
