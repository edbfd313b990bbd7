"""Host -> device cost of one iteration's inputs at configs[1] (what a PCIe-inclusive rate would add to bench.py's
HBM-resident number).  Two ways a training loop can hand over the two batches (real molecules, generator input):
  dense : the padded one-hot tensors A [B,N,N,E], X [B,N,M] in float32 from pinned host memory;
  coo   : the reference loader's form (utils.py:128-142): COO edge_index / edge_attr + node features, densified on
          the GPU by dg_densify (druggen_amd/data.py).
usage: python scripts/h2d_probe.py [B]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from druggen_amd import synth
from druggen_amd.data import dense_one_hot_adjacency

B, N, E, M = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 45, 5, 13
dev = torch.device("cuda:0")
a, x, bonds, atoms = synth.molecule_batch(B, N, E, M, seed=1)
ah, xh = torch.from_numpy(a).pin_memory(), torch.from_numpy(x).pin_memory()
src, dst = np.nonzero(bonds.reshape(B * N, N))          # row = b * N + i, col = j
ei = torch.from_numpy(np.stack([src, (src // N) * N + dst])).pin_memory()
ea = torch.from_numpy(bonds.reshape(B * N, N)[src, dst]).pin_memory()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def dense():
    return ah.to(dev, non_blocking=True), xh.to(dev, non_blocking=True)


def coo():
    e, t = ei.to(dev, non_blocking=True), ea.to(dev, non_blocking=True)
    return dense_one_hot_adjacency(e, t, B, N, E, check=False), xh.to(dev, non_blocking=True)


d_ms, c_ms = timed(dense), timed(coo)
print(f"B={B}: dense {ah.numel() * 4 / 1e6 + xh.numel() * 4 / 1e6:.1f} MB per batch: {d_ms:.3f} ms ; "
      f"coo {ei.numel() * 8 / 1e6 + ea.numel() * 8 / 1e6 + xh.numel() * 4 / 1e6:.2f} MB + densify: {c_ms:.3f} ms  (x2 batches per step)")
