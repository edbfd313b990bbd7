#!/usr/bin/env python
"""bf16 configuration against the fp32 HIP path at REAL batch sizes ("weak" 1): the golden batches hold 2-4
molecules, where single roundings dominate small-norm gradient tensors.  Same weights (PyTorch default init, seed 0),
same synthetic molecule batch and eps, one D step + one G step (train.py:351-384 without the optimizer updates) in both
activation modes; per-tensor error of every parameter gradient with the error model of the fp32 tests
(||got - want|| / max(||want||, ||all grads|| / sqrt(n_tensors)), SURVEY.md section 7 hard part 4).

    python scripts/bf16_vs_f32_probe.py [B ...]          (default: 32 256)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from druggen_amd import functional as dgf, synth
from druggen_amd.model import Discriminator, Generator, discriminator_loss, generator_loss


def run(B, mode, N=45, E=5, M=13, L=4):
    dev = torch.device("cuda")
    ctor = ("relu", N, E, M, 0.0)
    kw = dict(dim=128, depth=L, heads=8, mlp_ratio=3)
    torch.manual_seed(0)
    G, D = Generator(*ctor, **kw).to(dev), Discriminator(*ctor, **kw).to(dev)
    a, x, _, _ = synth.molecule_batch(B, N, E, M, seed=1234)
    da, dx, _, _ = synth.molecule_batch(B, N, E, M, seed=2234)
    ee, en = synth.interpolation_eps(B, 1234)
    t = lambda v: torch.from_numpy(v).to(dev)
    ge, gn, de, dn = t(a), t(x), t(da), t(dx)
    eps = (t(ee), t(en))
    out = {}
    with dgf.activations(mode):
        _, _, d_loss = discriminator_loss(G, D, de, dn, ge, gn, B, dev, 10.0, eps=eps)
        d_loss.backward()
        out["d_loss"] = float(d_loss)
        out["D"] = {k: None if p.grad is None else p.grad.detach().double().cpu().numpy() for k, p in D.named_parameters()}
        for p in list(G.parameters()) + list(D.parameters()):
            p.grad = None
        g_loss = generator_loss(G, D, ge, gn, B)[0]
        g_loss.backward()
        out["g_loss"] = float(g_loss)
        out["G"] = {k: None if p.grad is None else p.grad.detach().double().cpu().numpy() for k, p in G.named_parameters()}
    return out


def table_errors(got, want):
    names = [k for k, v in want.items() if v is not None]
    total = np.sqrt(sum(float((want[k] ** 2).sum()) for k in names))
    floor = total / np.sqrt(len(names))
    errs = []
    for k in names:
        err = np.linalg.norm((got[k] - want[k]).ravel())
        errs.append((err / max(np.linalg.norm(want[k].ravel()), floor), k))
    glob = np.sqrt(sum(float(((got[k] - want[k]) ** 2).sum()) for k in names)) / total
    return sorted(errs, reverse=True), glob


def main():
    batches = [int(v) for v in sys.argv[1:]] or [32, 256]
    for B in batches:
        ref = run(B, torch.float32)
        low = run(B, torch.bfloat16)
        print(f"B={B}: d_loss fp32 {ref['d_loss']:.6f} bf16 {low['d_loss']:.6f} (rel {abs(low['d_loss'] - ref['d_loss']) / max(1, abs(ref['d_loss'])):.2e})"
              f" | g_loss fp32 {ref['g_loss']:.6f} bf16 {low['g_loss']:.6f} (rel {abs(low['g_loss'] - ref['g_loss']) / max(1, abs(ref['g_loss'])):.2e})")
        for net in ("D", "G"):
            errs, glob = table_errors(low[net], ref[net])
            e = np.array([v for v, _ in errs])
            q = np.quantile(e, [0.5, 0.9, 0.99])
            print(f"  {net} gradients, {len(e)} tensors: global (all tensors as one vector) {glob:.4f} | per tensor median {q[0]:.4f} "
                  f"p90 {q[1]:.4f} p99 {q[2]:.4f} worst {e[0]:.4f}")
            over = [(v, k) for v, k in errs if v > 0.05]
            print(f"    above 5 %: {len(over)}" + "".join(f"\n      {v:.3f}  {k}" for v, k in over[:12]))


if __name__ == "__main__":
    main()
