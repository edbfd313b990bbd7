"""Case table and input builders shared by ``make_golden.py`` (which runs the
reference in the build container) and the parity tests (which run the oracle
and the HIP path).  Everything here is regenerated from seeds with numpy, so the
committed ``*.npz`` fixtures only carry the reference's *outputs*.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from druggen_amd import synth  # noqa: E402
from oracle.druggen_oracle import NetConfig, discriminator_schema, generator_schema  # noqa: E402

# name -> dict(cfg=NetConfig kwargs, batch, submodel, seed, lambda_gp, full)
# ``full`` fixtures keep every gradient tensor; the others keep per-tensor
# norms and random projections (see ``summarise``).
CASES = {
    "tiny_relu": dict(cfg=dict(act="relu", vertexes=6, edges=4, nodes=5, dim=16, depth=2, heads=4, mlp_ratio=3),
                      batch=3, submodel="DrugGEN", seed=11, lambda_gp=10.0, full=True),
    "tiny_tanh": dict(cfg=dict(act="tanh", vertexes=5, edges=3, nodes=4, dim=16, depth=1, heads=2, mlp_ratio=2),
                      batch=2, submodel="NoTarget", seed=12, lambda_gp=10.0, full=True),
    "tiny_leaky": dict(cfg=dict(act="leaky", vertexes=7, edges=5, nodes=6, dim=16, depth=2, heads=8, mlp_ratio=3),
                       batch=2, submodel="DrugGEN", seed=13, lambda_gp=5.0, full=True),
    "tiny_sigmoid": dict(cfg=dict(act="sigmoid", vertexes=4, edges=2, nodes=3, dim=8, depth=3, heads=1, mlp_ratio=4),
                         batch=4, submodel="NoTarget", seed=14, lambda_gp=10.0, full=True),
    # BASELINE.json configs[0] shape (N=9, L=1, dim 128) at reduced batch
    "c1_b4": dict(cfg=dict(act="relu", vertexes=9, edges=5, nodes=5, dim=128, depth=1, heads=8, mlp_ratio=3),
                  batch=4, submodel="NoTarget", seed=21, lambda_gp=10.0, full=False),
    # BASELINE.json configs[1] shape (N=45, L=4, E=5, M=13, dim 128) at batch 2
    "c2_b2": dict(cfg=dict(act="relu", vertexes=45, edges=5, nodes=13, dim=128, depth=4, heads=8, mlp_ratio=3),
                  batch=2, submodel="DrugGEN", seed=22, lambda_gp=10.0, full=False),
    # dim-128 (fused-kernel) path with a non-ReLU embedding / head activation
    "c1_tanh_b4": dict(cfg=dict(act="tanh", vertexes=9, edges=5, nodes=5, dim=128, depth=1, heads=8, mlp_ratio=3),
                       batch=4, submodel="DrugGEN", seed=24, lambda_gp=10.0, full=False),
    # BASELINE.json configs[4] geometry (N=90 atoms, 10 bond types) at depth 2, batch 2
    "c5_b2": dict(cfg=dict(act="relu", vertexes=90, edges=10, nodes=13, dim=128, depth=2, heads=8, mlp_ratio=3),
                  batch=2, submodel="NoTarget", seed=25, lambda_gp=10.0, full=False),
    # Real molecular graphs: SMILES shipped with the reference's results (tests/golden/chembl_like_smiles.csv),
    # featurised by druggen_amd.smiles with the atom / bond tables the reference's encoder construction
    # (src/data/utils.py:70-126) yields over those result files: atoms {PAD,B,C,N,O,F,P,S,Cl}, bonds
    # {ZERO,SINGLE,DOUBLE,TRIPLE,AROMATIC}; max_atom 45, default 4-layer / 8-head network.
    "chembl_b4": dict(cfg=dict(act="relu", vertexes=45, edges=5, nodes=9, dim=128, depth=4, heads=8, mlp_ratio=3),
                      batch=4, submodel="DrugGEN", seed=23, lambda_gp=10.0, full=False, source="smiles"),
}

CHEMBL_ATOM_ENCODER = {0: 0, 5: 1, 6: 2, 7: 3, 8: 4, 9: 5, 15: 6, 16: 7, 17: 8}
CHEMBL_BOND_ENCODER = {0: 0, 1: 1, 2: 2, 3: 3, 12: 4}


def smiles_batches(case: dict):
    """(generator-side, drug-side) dense one-hot batches of the real-molecule sample."""
    from druggen_amd import smiles as sm
    cfg = net_config(case)
    rows = [ln.strip().split(",") for ln in open(os.path.join(HERE, "chembl_like_smiles.csv"))
            if ln.strip() and not ln.startswith("#")][1:]
    out = []
    for role in ("mol", "drug"):
        graphs = [sm.molecule_graph(r[2], CHEMBL_ATOM_ENCODER, CHEMBL_BOND_ENCODER, cfg.vertexes) for r in rows if r[0] == role]
        assert len(graphs) == case["batch"] and all(g is not None for g in graphs)
        a = np.zeros((len(graphs), cfg.vertexes, cfg.vertexes, cfg.edges), dtype=np.float32)
        for b, g in enumerate(graphs):
            labels = np.zeros((cfg.vertexes, cfg.vertexes), dtype=np.int64)
            labels[g.edge_index[0], g.edge_index[1]] = g.edge_attr
            a[b] = np.eye(cfg.edges, dtype=np.float32)[labels]
        out.append((a, np.stack([g.x for g in graphs])))
    return out


def net_config(case: dict) -> NetConfig:
    return NetConfig(dropout=0.0, **case["cfg"])


def build_inputs(case: dict):
    """Numpy inputs of one GAN step: generator batch, D-real batch, eps."""
    cfg = net_config(case)
    B, seed = case["batch"], case["seed"]
    if case.get("source") == "smiles":
        (a, x), (da, dx) = smiles_batches(case)
        eps_edge, eps_node = synth.interpolation_eps(B, seed)
        return dict(gen_edge=a, gen_node=x, disc_edge=da, disc_node=dx, eps_edge=eps_edge, eps_node=eps_node)
    a, x, _, _ = synth.molecule_batch(B, cfg.vertexes, cfg.edges, cfg.nodes, seed=1000 + seed)
    if case["submodel"] == "DrugGEN":       # train.py:340-342: independent drug batch
        da, dx, _, _ = synth.molecule_batch(B, cfg.vertexes, cfg.edges, cfg.nodes, seed=2000 + seed)
    else:                                   # train.py:343-345: NoTarget reuses the batch
        da, dx = a, x
    eps_edge, eps_node = synth.interpolation_eps(B, seed)
    return dict(gen_edge=a, gen_node=x, disc_edge=da, disc_node=dx, eps_edge=eps_edge, eps_node=eps_node)


def build_params(case: dict):
    """Deterministic (G, D) weights keyed by reference state_dict names."""
    cfg = net_config(case)
    g = synth.fill_parameters(generator_schema(cfg), seed=case["seed"] * 10 + 1, gain=1.7)
    d = synth.fill_parameters(discriminator_schema(cfg), seed=case["seed"] * 10 + 2, gain=1.7)
    return g, d


def probes(shape, index: int, k: int = 4) -> np.ndarray:
    """k fixed random +-1/uniform probe tensors for gradient summaries."""
    rng = np.random.Generator(np.random.PCG64([4242, index]))
    return rng.random((k,) + tuple(shape)) * 2.0 - 1.0


def summarise(arr: np.ndarray, index: int) -> np.ndarray:
    """[l2 norm, 4 probe projections, first element, last element] in float64."""
    a = np.asarray(arr, dtype=np.float64)
    pr = probes(a.shape, index)
    proj = (pr * a[None]).reshape(pr.shape[0], -1).sum(1)
    flat = a.reshape(-1)
    return np.concatenate([[np.sqrt((flat ** 2).sum())], proj, [flat[0], flat[-1]]])


def fixture_path(name: str) -> str:
    return os.path.join(HERE, name + ".npz")
