"""CPU-side checks: the C-ABI library builds/loads and exports every declared
symbol, the Python boundary mirrors the reference API, the product path refuses
to run without a GPU, and the data-parallel step is exact across 2 gloo ranks."""
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import harness
from oracle import druggen_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from druggen_amd import _lib, build
    build.build()
    header = open(os.path.join(ROOT, "include", "druggen_hip.h")).read()
    declared = set(re.findall(r"\b(dg_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dg_version() == 232
    assert lib.dg_last_error_string() is not None


def test_argument_validation_needs_no_gpu():
    from druggen_amd import _lib
    lib = _lib.load()
    assert lib.dg_attn_core_fwd(None, None, None, None, None, None, 1, 9, 128, 0.25, 0, None) == -2
    assert b"null pointer" in lib.dg_last_error_string()
    assert lib.dg_ln_workspace_bytes(518400, 128) > 0
    assert lib.dg_ln_workspace_bytes(10, 6) == 0           # C % 4 != 0 -> unsupported


def test_product_path_fails_loudly_without_gpu():
    from druggen_amd import functional as dgf
    from druggen_amd.model import Generator
    G = Generator("relu", 6, 4, 5, 0.0, dim=16, depth=1, heads=4, mlp_ratio=3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G(torch.zeros(1, 6, 6, 4), torch.zeros(1, 6, 5))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            dgf.ln_residual(torch.zeros(4, 16), None, torch.ones(16), torch.zeros(16))
        with pytest.raises(RuntimeError):
            dgf.attn_core(torch.zeros(1, 3, 8), torch.zeros(1, 3, 8), torch.zeros(1, 3, 8), torch.zeros(1, 3, 3, 8), 0.5)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "druggen_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert "oracle" not in src, os.path.join(base, f)
    for f in ("layers.py", "models.py", "loss.py"):
        assert "oracle" not in open(os.path.join(ROOT, "src", "model", f)).read()


def test_module_api_mirrors_reference_signatures():
    """Constructor / forward signatures of src/model/{layers,models,loss}.py."""
    import inspect
    from src.model import layers, loss, models
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(models.Generator.__init__)[1:] == ["act", "vertexes", "edges", "nodes", "dropout", "dim", "depth", "heads", "mlp_ratio"]
    assert sig(models.Discriminator.__init__)[1:] == ["act", "vertexes", "edges", "nodes", "dropout", "dim", "depth", "heads", "mlp_ratio"]
    assert sig(models.Generator.forward)[1:] == ["z_e", "z_n"] and sig(models.Discriminator.forward)[1:] == ["z_e", "z_n"]
    assert sig(models.simple_disc.__init__)[1:] == ["act", "m_dim", "vertexes", "b_dim"]
    assert sig(layers.TransformerEncoder.__init__)[1:] == ["dim", "depth", "heads", "act", "mlp_ratio", "drop_rate"]
    assert sig(layers.Encoder_Block.__init__)[1:] == ["dim", "heads", "act", "mlp_ratio", "drop_rate"]
    assert sig(layers.MHA.__init__)[1:] == ["dim", "heads", "attention_dropout"]
    assert sig(layers.MLP.__init__)[1:] == ["in_feat", "hid_feat", "out_feat", "dropout"]
    assert sig(loss.gradient_penalty)[:7] == ["discriminator", "real_node", "real_edge", "fake_node", "fake_edge", "batch_size", "device"]
    assert sig(loss.discriminator_loss)[:9] == ["generator", "discriminator", "drug_adj", "drug_annot", "mol_adj", "mol_annot", "batch_size", "device", "lambda_gp"]
    assert sig(loss.generator_loss)[:5] == ["generator", "discriminator", "mol_adj", "mol_annot", "batch_size"]
    blk = layers.Encoder_Block(16, 4, None, 3, 0.0)
    assert [n for n, _ in blk.named_children()] == ["ln1", "attn", "ln3", "ln4", "mlp", "mlp2", "ln5", "ln6"]


def test_synthetic_graphs_are_valid_one_hot_molecules():
    from druggen_amd import synth
    a, x, bonds, atoms = synth.molecule_batch(16, 45, 5, 13, seed=3)
    assert a.shape == (16, 45, 45, 5) and x.shape == (16, 45, 13) and a.dtype == np.float32
    assert np.all(a.sum(-1) == 1) and np.all(x.sum(-1) == 1)
    assert np.all(bonds == bonds.transpose(0, 2, 1)) and np.all(np.diagonal(bonds, axis1=1, axis2=2) == 0)
    n_atoms = (atoms > 0).sum(1)
    assert n_atoms.min() >= 15 and n_atoms.max() <= 45
    frac_no_bond = (bonds == 0).mean()
    assert 0.9 < frac_no_bond < 0.995
    a2, *_ = synth.molecule_batch(16, 45, 5, 13, seed=3)
    assert np.array_equal(a, a2)


# ---------------------------------------------------------------------------
# data-parallel step: 2 gloo ranks == 1 rank on the full batch
# ---------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_losses():
    def d_loss_fn(G, D, de, dn, ge, gn, B, dev, lam, eps=None):
        return orc.discriminator_loss(G, D, de, dn, ge, gn, lam, eps[0], eps[1])

    def g_loss_fn(G, D, ge, gn, B):
        return orc.generator_loss(G, D, ge, gn)
    return d_loss_fn, g_loss_fn


def _make_nets(case, dtype):
    cfg = cases.net_config(case)
    gp, dp = cases.build_params(case)
    G = orc.OracleNet("G", cfg, {k: torch.from_numpy(v).to(dtype) for k, v in gp.items()})
    D = orc.OracleNet("D", cfg, {k: torch.from_numpy(v).to(dtype) for k, v in dp.items()})
    return G, D


def _dp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from druggen_amd.trainer import GANStep, broadcast_parameters
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    case = dict(cases.CASES["tiny_relu"], batch=4)
    dtype = torch.float64
    G, D = _make_nets(case, dtype)
    if rank != 0:                       # prove the initial broadcast matters
        with torch.no_grad():
            for p in list(G.parameters()) + list(D.parameters()):
                p.add_(0.5)
    broadcast_parameters(G)
    broadcast_parameters(D)
    inp = harness.torch_inputs(case, dtype)
    per = case["batch"] // world
    sl = slice(rank * per, (rank + 1) * per)
    d_fn, g_fn = _oracle_losses()
    stepper = GANStep(G, D, lambda_gp=case["lambda_gp"], d_loss_fn=d_fn, g_loss_fn=g_fn,
                      skip_d_wgrad_in_g_step=False)
    for _ in range(2):
        stepper.step(inp["disc_edge"][sl], inp["disc_node"][sl], inp["gen_edge"][sl], inp["gen_node"][sl],
                     eps=(inp["eps_edge"][sl], inp["eps_node"][sl]))
    torch.save({"G": [p.detach() for p in G.parameters()], "D": [p.detach() for p in D.parameters()],
                "D_none": [p.grad is None for p in D.parameters()]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gloo_step_equals_single_rank_full_batch(tmp_path):
    from druggen_amd.trainer import GANStep
    world, port = 2, _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    for a, b in zip(r0["G"] + r0["D"], r1["G"] + r1["D"]):
        assert torch.equal(a, b), "ranks diverged"
    # single process, full batch
    case = dict(cases.CASES["tiny_relu"], batch=4)
    G, D = _make_nets(case, torch.float64)
    inp = harness.torch_inputs(case, torch.float64)
    d_fn, g_fn = _oracle_losses()
    stepper = GANStep(G, D, lambda_gp=case["lambda_gp"], d_loss_fn=d_fn, g_loss_fn=g_fn,
                      skip_d_wgrad_in_g_step=False)
    for _ in range(2):
        stepper.step(inp["disc_edge"], inp["disc_node"], inp["gen_edge"], inp["gen_node"],
                     eps=(inp["eps_edge"], inp["eps_node"]))
    for a, b in zip(r0["G"] + r0["D"], list(G.parameters()) + list(D.parameters())):
        assert torch.allclose(a, b.detach(), rtol=0, atol=1e-9), (a - b).abs().max()
    # dead discriminator parameters stayed grad-less on the ranks too
    assert r0["D_none"] == [p.grad is None for p in D.parameters()]
    assert sum(r0["D_none"]) == 10


def test_grad_bucket_skips_none_grads_single_process():
    from druggen_amd.trainer import GradBucket
    lin = torch.nn.Linear(3, 2)
    bucket = GradBucket(lin)
    bucket.all_reduce_mean()            # world size 1: no-op, no error
    assert lin.weight.grad is None


def test_checkpoint_roundtrip_and_dataparallel_prefix(tmp_path):
    """train.py:250-263 / inference.py:135-139 file names, bare state_dicts, `module.` prefix."""
    from druggen_amd import checkpoint
    from druggen_amd.model import Discriminator, Generator
    args = ("relu", 6, 4, 5, 0.0)
    kw = dict(dim=16, depth=1, heads=4, mlp_ratio=2)
    G, D = Generator(*args, **kw), Discriminator(*args, **kw)
    checkpoint.save_model(G, D, str(tmp_path), 2, 99)
    assert sorted(os.listdir(tmp_path)) == ["3-100-D.ckpt", "3-100-G.ckpt"]
    G2, D2 = Generator(*args, **kw), Discriminator(*args, **kw)
    checkpoint.restore_model(G2, D2, str(tmp_path), 3, 100)
    for a, b in zip(list(G.state_dict().values()) + list(D.state_dict().values()),
                    list(G2.state_dict().values()) + list(D2.state_dict().values())):
        assert torch.equal(a, b)
    # a checkpoint written from nn.DataParallel (train.py:262 saves the wrapper's state_dict)
    torch.save({"module." + k: v for k, v in G.state_dict().items()}, tmp_path / "DrugGEN-G.ckpt")
    G3 = Generator(*args, **kw)
    checkpoint.load_generator(G3, str(tmp_path), "DrugGEN")
    assert all(torch.equal(a, b) for a, b in zip(G.state_dict().values(), G3.state_dict().values()))


def test_aux_oracle_to_dense_adj_semantics():
    """The restated PyG to_dense_adj: scatter-add, per-graph local indices, zero padding."""
    from oracle import aux_oracle as aux
    ei = np.array([[0, 1, 4, 4], [1, 0, 5, 5]])
    adj = aux.to_dense_adj(ei, np.array([0, 0, 0, 1, 1, 1]), np.array([2, 2, 1, 1]), 3)
    assert adj.shape == (2, 3, 3) and adj[0, 0, 1] == 2 and adj[0, 1, 0] == 2 and adj[1, 1, 2] == 2 and adj.sum() == 6
    oh = aux.label2onehot(adj, 4)
    assert oh.shape == (2, 3, 3, 4) and np.all(oh.sum(-1) == 1) and oh[1, 1, 2, 2] == 1


# PyG's own published vectors for to_dense_adj: the "Examples" block of its docstring (torch_geometric/utils/to_dense_adj.py, the
# 2.2 series the reference pins: environment.yml:171 `pyg=2.2.0`) -- data, not code.  They pin the oracle's restatement (and, in
# tests/test_hip_aux.py, dg_densify) to the package's documented behaviour, including an edge (3 -> 0) that leaves its graph.
PYG_EDGE_INDEX = [[0, 0, 1, 2, 3], [0, 1, 0, 3, 0]]
PYG_BATCH = [0, 0, 1, 1]
PYG_DENSE = [[[1, 1], [1, 0]], [[0, 1], [1, 0]]]                                   # to_dense_adj(edge_index, batch)
PYG_DENSE_MAX4 = [[[1, 1, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]],        # ... max_num_nodes=4
                  [[0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]]]
PYG_EDGE_ATTR = [1, 2, 3, 4, 5]
PYG_DENSE_ATTR = [[[1, 2], [3, 0]], [[0, 4], [5, 0]]]                               # to_dense_adj(edge_index, batch, edge_attr)


def test_aux_oracle_reproduces_pyg_docstring_examples():
    from oracle import aux_oracle as aux
    ei, batch = np.array(PYG_EDGE_INDEX), np.array(PYG_BATCH)
    ones = np.ones(5, dtype=np.int64)
    assert aux.to_dense_adj(ei, batch, ones, 2).tolist() == PYG_DENSE
    assert aux.to_dense_adj(ei, batch, ones, 4).tolist() == PYG_DENSE_MAX4
    assert aux.to_dense_adj(ei, batch, np.array(PYG_EDGE_ATTR), 2).tolist() == PYG_DENSE_ATTR
    # single graph (batch of zeros), the first example's first graph
    assert aux.to_dense_adj(np.array([[0, 0, 1], [0, 1, 0]]), np.zeros(2, dtype=np.int64), np.ones(3, dtype=np.int64), 2).tolist() == [PYG_DENSE[0]]


# ---- SMILES -> graph (druggen_amd.smiles; reference src/data/dataset.py:119-160,280-316, utils.py:70-126) ----
def test_smiles_parser_known_molecules():
    from druggen_amd import smiles as sm
    atoms, arom, bonds = sm.parse_smiles("c1ccccc1")
    assert atoms == [6] * 6 and all(arom) and len(bonds) == 6 and {t for _, _, t in bonds} == {sm.AROMATIC}
    atoms, _, bonds = sm.parse_smiles("CC(=O)O")
    assert atoms == [6, 6, 8, 8] and sorted(bonds) == [(0, 1, sm.SINGLE), (1, 2, sm.DOUBLE), (1, 3, sm.SINGLE)]
    # biphenyl: the inter-ring bond is written '-' and stays single; nitrile triple bond; %nn closure
    _, _, bonds = sm.parse_smiles("c1ccccc1-c1ccccc1")
    assert sum(t == sm.SINGLE for _, _, t in bonds) == 1 and sum(t == sm.AROMATIC for _, _, t in bonds) == 12
    assert (0, 1, sm.TRIPLE) in sm.parse_smiles("N#CC")[2]
    assert len(sm.parse_smiles("C%12CCCCC%12")[2]) == 6
    # bracket atoms: charge, H count, chirality; two-letter halogens; ring bond given at the closing digit
    atoms, arom, _ = sm.parse_smiles("[NH3+]C[C@H](Cl)c1cc[nH]c1Br")
    assert atoms == [7, 6, 6, 17, 6, 6, 6, 7, 6, 35] and arom[7] and not arom[0]
    assert (0, 5, sm.DOUBLE) in sm.parse_smiles("C1CCCCC=1")[2]
    for bad in ("C(", "C1CC", "C)", "[Xx]", "C.C", "[H]C", ""):
        with pytest.raises(sm.SmilesError):
            sm.parse_smiles(bad)


def test_smiles_encoders_and_graph_layout():
    from druggen_amd import smiles as sm
    pool = ["CCO", "c1ccccc1Cl", "N#CC", "not a smiles", "C" * 50]
    atom_enc, atom_dec, bond_enc, bond_dec, kept, max_len = sm.build_encoders(pool, max_atom=45)
    assert kept == pool[:3] and max_len == 7
    assert atom_enc == {0: 0, 6: 1, 7: 2, 8: 3, 17: 4} and atom_dec[4] == 17        # PAD first, then sorted Z
    assert bond_enc == {0: 0, sm.SINGLE: 1, sm.TRIPLE: 2, sm.AROMATIC: 3}            # no DOUBLE seen -> labels shift
    g = sm.molecule_graph("CCO", atom_enc, bond_enc, 5)
    assert g.num_atoms == 3 and g.x.shape == (5, 5)
    assert g.x.argmax(-1).tolist() == [1, 1, 3, 0, 0]                                 # padded with PAD rows
    assert g.edge_index.tolist() == [[0, 1, 1, 2], [1, 0, 2, 1]] and g.edge_attr.tolist() == [1, 1, 1, 1]
    assert sm.molecule_graph("C=C", atom_enc, bond_enc, 5) is None                    # bond type not in the encoder
    assert sm.molecule_graph("CCCCCC", atom_enc, bond_enc, 5) is None                 # too many atoms
    assert sm.molecule_graph("C", atom_enc, bond_enc, 5) is None                      # an atom without bonds
    batch = sm.collate([g, sm.molecule_graph("N#CC", atom_enc, bond_enc, 5)])
    assert batch.x.shape == (10, 5) and batch.batch.tolist() == [0] * 5 + [1] * 5
    assert batch.edge_index[:, 4:].tolist() == [[5, 6, 6, 7], [6, 5, 7, 6]] and batch.edge_attr.tolist()[4:] == [2, 2, 1, 1]


def test_smiles_sample_matches_oracle_densify():
    """collate -> reference-style load_molecules (numpy restatement) gives the dense batches of the golden case."""
    import cases
    from druggen_amd import smiles as sm
    from oracle import aux_oracle
    case = cases.CASES["chembl_b4"]
    (a, x), _ = cases.smiles_batches(case)
    rows = [ln.strip().split(",") for ln in open(os.path.join(os.path.dirname(cases.__file__), "chembl_like_smiles.csv"))
            if ln.strip() and not ln.startswith("#")][1:]
    graphs = [sm.molecule_graph(r[2], cases.CHEMBL_ATOM_ENCODER, cases.CHEMBL_BOND_ENCODER, 45) for r in rows if r[0] == "mol"]
    batch = sm.collate(graphs)
    _, a2, x2 = aux_oracle.load_molecules(batch.edge_index.numpy(), batch.edge_attr.numpy(), batch.x.numpy(),
                                          batch.batch.numpy(), b_dim=5, batch_size=4)
    assert np.array_equal(a2, a) and np.array_equal(x2, x)
    assert [g.num_atoms for g in graphs] == [39, 29, 29, 34]
    assert (a[..., 1:].sum((1, 2, 3)) == 2 * np.array([43, 33, 32, 38])).all()      # each bond appears twice
