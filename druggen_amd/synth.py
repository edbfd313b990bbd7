"""Seeded molecule-like graph batches and deterministic parameter fills.

Pure numpy, bit-reproducible across machines (PCG64 streams), so the same
call regenerates the golden-fixture inputs in this container and the bench /
parity inputs on the GPU box.  Shapes follow what the reference feeds its hot
path: ``a_tensor [B,N,N,E]`` one-hot f32 and ``x_tensor [B,N,M]`` one-hot f32
(reference ``src/data/utils.py:128-137``, SURVEY.md section 8d "Synthetic
inputs").
"""
from __future__ import annotations

import numpy as np

__all__ = ["molecule_batch", "fill_parameters", "interpolation_eps"]


def _rng(seed: int, stream: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([int(seed), int(stream)]))


def _bond_probabilities(n_bond_types: int) -> np.ndarray:
    """P(bond label) over labels 1..E-1 (label 0 = no bond)."""
    base = np.array([0.75, 0.10, 0.02, 0.13], dtype=np.float64)
    k = n_bond_types - 1
    if k <= 0:
        raise ValueError("need at least 2 bond classes (0 = no bond)")
    if k <= 4:
        p = base[:k].copy()
    else:
        p = np.concatenate([base * 0.9, np.full(k - 4, 0.1 / (k - 4))])
    return p / p.sum()


def molecule_batch(batch: int, vertexes: int, b_dim: int, m_dim: int, seed: int = 1234):
    """Return ``(a, x, bond_labels, atom_labels)`` for ``batch`` random molecules.

    Per molecule: n ~ U{ceil(N/3)..N} real atoms, atom labels 1..M-1 (0 = PAD
    beyond n); a random spanning tree over the n atoms plus floor(n/6) ring
    closures, bond labels 1..E-1, symmetric, zero diagonal.
    """
    rng = _rng(seed)
    N, E, M = int(vertexes), int(b_dim), int(m_dim)
    atoms = np.zeros((batch, N), dtype=np.int64)
    bonds = np.zeros((batch, N, N), dtype=np.int64)
    pb = _bond_probabilities(E)
    n_lo = max(2, -(-N // 3))
    for b in range(batch):
        n = int(rng.integers(n_lo, N + 1))
        if M > 1:
            atoms[b, :n] = rng.integers(1, M, size=n)
        order = rng.permutation(n)
        for t in range(1, n):
            u = int(order[t])
            w = int(order[int(rng.integers(0, t))])
            lab = 1 + int(rng.choice(E - 1, p=pb))
            bonds[b, u, w] = bonds[b, w, u] = lab
        for _ in range(n // 6):
            u, w = (int(v) for v in rng.integers(0, n, size=2))
            if u != w and bonds[b, u, w] == 0:
                lab = 1 + int(rng.choice(E - 1, p=pb))
                bonds[b, u, w] = bonds[b, w, u] = lab
    a = np.zeros((batch, N, N, E), dtype=np.float32)
    np.put_along_axis(a, bonds[..., None], 1.0, axis=-1)
    x = np.zeros((batch, N, M), dtype=np.float32)
    np.put_along_axis(x, atoms[..., None], 1.0, axis=-1)
    return a, x, bonds, atoms


def interpolation_eps(batch: int, seed: int):
    """(eps_edge [B,1,1,1], eps_node [B,1,1]) in the order the reference draws
    them (``src/model/loss.py:21-22``: edge first)."""
    rng = _rng(seed, 7)
    eps_edge = rng.random((batch, 1, 1, 1)).astype(np.float32)
    eps_node = rng.random((batch, 1, 1)).astype(np.float32)
    return eps_edge, eps_node


def fill_parameters(named_shapes, seed: int, gain: float = 1.0):
    """Deterministic weights for a ``state_dict`` schema.

    ``named_shapes``: iterable of ``(name, shape)`` in state_dict order.  Each
    tensor gets its own PCG64 stream (seed, index):  matrices ~ U(-k, k) with
    k = gain/sqrt(fan_in) (nn.Linear-like scale), LayerNorm weights near 1,
    biases small.  Returns ``{name: float32 ndarray}``.
    """
    out = {}
    for idx, (name, shape) in enumerate(named_shapes):
        rng = _rng(seed, 100 + idx)
        shape = tuple(int(s) for s in shape)
        u = rng.random(shape) * 2.0 - 1.0
        leaf = name.rsplit(".", 2)
        is_ln = any(part.startswith("ln") for part in name.split("."))
        if len(shape) >= 2:
            w = u * (gain / np.sqrt(shape[-1]))
        elif is_ln and leaf[-1] == "weight":
            w = 1.0 + 0.1 * u
        else:
            w = 0.1 * u
        out[name] = w.astype(np.float32)
    return out
