"""ORACLE -- test infrastructure only.  Never imported by ``druggen_amd``.

A CPU restatement (plain PyTorch ops, any float dtype) of the one DrugGEN hot
path this repo accelerates: the graph-transformer Generator / Discriminator
forward and the WGAN-GP losses.  It is written functionally over a flat
``{state_dict key: tensor}`` mapping so that reference checkpoints, the golden
fixtures under ``tests/golden`` and the HIP modules all share one schema.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file, and only as the checker / timed baseline.

Pinning: the reference ships no tests or golden vectors of its own
("parity unpinned" by reference-owned tests).  This oracle is
pinned instead against outputs of the reference itself, produced in the build
container by ``tests/golden/make_golden.py`` (which imports
``/root/reference/src/model``) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` replays them.

Every function cites the reference lines it restates (paths relative to the
reference repository root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Callable, Dict, List, Mapping, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Mapping[str, torch.Tensor]


@dataclass(frozen=True)
class NetConfig:
    """Constructor arguments of Generator / Discriminator
    (src/model/models.py:13, :115)."""
    act: str = "relu"
    vertexes: int = 9
    edges: int = 5       # b_dim, bond classes
    nodes: int = 5       # m_dim, atom classes
    dropout: float = 0.0
    dim: int = 128
    depth: int = 1
    heads: int = 8
    mlp_ratio: int = 3

    def as_kwargs(self):
        return asdict(self)


# --------------------------------------------------------------------------
# schema
# --------------------------------------------------------------------------
def _linear(prefix: str, fan_out: int, fan_in: int):
    return [(prefix + ".weight", (fan_out, fan_in)), (prefix + ".bias", (fan_out,))]


def _norm(prefix: str, dim: int):
    return [(prefix + ".weight", (dim,)), (prefix + ".bias", (dim,))]


def _block_schema(prefix: str, cfg: NetConfig):
    """Parameter order of Encoder_Block.__init__ (src/model/layers.py:165-172)
    with MHA's q,k,v,e,out_e,out_n (layers.py:86-95) and MLP's fc1,fc2 (:36-38)."""
    C, H = cfg.dim, cfg.dim * cfg.mlp_ratio
    s = _norm(prefix + "ln1", C)
    for name in ("q", "k", "v", "e", "out_e", "out_n"):
        s += _linear(prefix + "attn." + name, C, C)
    s += _norm(prefix + "ln3", C) + _norm(prefix + "ln4", C)
    for m in ("mlp", "mlp2"):
        s += _linear(prefix + m + ".fc1", H, C) + _linear(prefix + m + ".fc2", C, H)
    s += _norm(prefix + "ln5", C) + _norm(prefix + "ln6", C)
    return s


def _trunk_schema(cfg: NetConfig):
    """node_layers / edge_layers / TransformerEncoder (models.py:52-65, 154-168)."""
    s = _linear("node_layers.0", 64, cfg.nodes) + _linear("node_layers.2", cfg.dim, 64)
    s += _linear("edge_layers.0", 64, cfg.edges) + _linear("edge_layers.2", cfg.dim, 64)
    for l in range(cfg.depth):
        s += _block_schema(f"TransformerEncoder.Encoder_Blocks.{l}.", cfg)
    return s


def generator_schema(cfg: NetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """state_dict keys/shapes of Generator (models.py:52-68)."""
    return _trunk_schema(cfg) + _linear("readout_e", cfg.edges, cfg.dim) + _linear("readout_n", cfg.nodes, cfg.dim)


def discriminator_schema(cfg: NetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """state_dict keys/shapes of Discriminator (models.py:154-178)."""
    s = _trunk_schema(cfg)
    widths = [cfg.vertexes * cfg.dim, 64, 32, 16, 1]
    for i in range(4):
        s += _linear(f"node_mlp.{2 * i}", widths[i + 1], widths[i])
    return s


def discriminator_dead_parameters(cfg: NetConfig) -> List[str]:
    """Keys that receive no gradient in the reference: the edge branch of the
    last encoder block feeds nothing in Discriminator.forward (models.py:202-207)."""
    p = f"TransformerEncoder.Encoder_Blocks.{cfg.depth - 1}."
    dead = []
    for mod in ("attn.out_e", "ln4", "mlp2.fc1", "mlp2.fc2", "ln6"):
        dead += [p + mod + ".weight", p + mod + ".bias"]
    return dead


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
def activation(name: str) -> Callable[[torch.Tensor], torch.Tensor]:
    """String -> activation (models.py:39-46); LeakyReLU default slope 0.01."""
    table = {
        "relu": torch.relu,
        "leaky": lambda t: F.leaky_relu(t, 0.01),
        "sigmoid": torch.sigmoid,
        "tanh": torch.tanh,
    }
    return table[name]


def _lin(P: Params, key: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, P[key + ".weight"], P[key + ".bias"])


def _ln(P: Params, key: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[key + ".weight"], P[key + ".bias"], 1e-5)


def _drop(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    return F.dropout(x, p, training) if (p > 0.0 and training) else x


def mlp(P: Params, key: str, x: torch.Tensor, p_drop: float = 0.0, training: bool = False):
    """MLP.forward (layers.py:40-54): fc2(relu(fc1 x)), always ReLU."""
    return _drop(_lin(P, key + ".fc2", torch.relu(_lin(P, key + ".fc1", x))), p_drop, training)


def attention_scores(q, k, e, heads: int):
    """layers.py:119-125.  s[b,i,j,c] = q[b,i,c] k[b,j,c] / sqrt(C/H) * (e+1) e.
    Per channel -- there is no contraction over the head dimension."""
    d_k = q.shape[-1] // heads
    s = q.unsqueeze(2) * k.unsqueeze(1)
    s = s / math.sqrt(d_k)
    return s * (e + 1) * e


def attention_aggregate(s, v):
    """layers.py:130-134: softmax over the neighbour axis j, then sum_j p v_j."""
    p = torch.softmax(s, dim=2)
    return (p * v.unsqueeze(1)).sum(dim=2)


def mha(P: Params, key: str, node, edge, heads: int):
    """MHA.forward (layers.py:97-137) -> (node_out, edge_out)."""
    q = _lin(P, key + ".q", node)
    k = _lin(P, key + ".k", node)
    v = _lin(P, key + ".v", node)
    e = _lin(P, key + ".e", edge)
    s = attention_scores(q, k, e, heads)
    edge_out = _lin(P, key + ".out_e", s)
    node_out = _lin(P, key + ".out_n", attention_aggregate(s, v))
    return node_out, edge_out


def encoder_block(P: Params, key: str, x, y, cfg: NetConfig, training: bool = False):
    """Encoder_Block.forward (layers.py:174-193).  The node residual uses the
    normalised input x1 (there is no ln2)."""
    x1 = _ln(P, key + "ln1", x)
    x2, y1 = mha(P, key + "attn", x1, y, cfg.heads)
    x2 = _ln(P, key + "ln3", x1 + x2)
    y2 = _ln(P, key + "ln4", y + y1)
    x = _ln(P, key + "ln5", x2 + mlp(P, key + "mlp", x2, cfg.dropout, training))
    y = _ln(P, key + "ln6", y2 + mlp(P, key + "mlp2", y2, cfg.dropout, training))
    return x, y


def transformer_encoder(P: Params, x, y, cfg: NetConfig, training: bool = False):
    """TransformerEncoder.forward (layers.py:220-234)."""
    for l in range(cfg.depth):
        x, y = encoder_block(P, f"TransformerEncoder.Encoder_Blocks.{l}.", x, y, cfg, training)
    return x, y


def embed(P: Params, z_e, z_n, cfg: NetConfig, training: bool = False):
    """node_layers / edge_layers + symmetrise (models.py:91-94, 196-199)."""
    act = activation(cfg.act)
    node = _drop(act(_lin(P, "node_layers.2", act(_lin(P, "node_layers.0", z_n)))), cfg.dropout, training)
    edge = _drop(act(_lin(P, "edge_layers.2", act(_lin(P, "edge_layers.0", z_e)))), cfg.dropout, training)
    edge = (edge + edge.permute(0, 2, 1, 3)) / 2
    return node, edge


def generator_forward(P: Params, z_e, z_n, cfg: NetConfig, training: bool = False):
    """Generator.forward (models.py:71-103): edge argument first.
    Returns (node, edge, node_sample logits, edge_sample logits)."""
    node, edge = embed(P, z_e, z_n, cfg, training)
    node, edge = transformer_encoder(P, node, edge, cfg, training)
    return node, edge, _lin(P, "readout_n", node), _lin(P, "readout_e", edge)


def discriminator_forward(P: Params, z_e, z_n, cfg: NetConfig, training: bool = False):
    """Discriminator.forward (models.py:180-209) -> logits [B,1]."""
    act = activation(cfg.act)
    node, edge = embed(P, z_e, z_n, cfg, training)
    node, _ = transformer_encoder(P, node, edge, cfg, training)
    h = node.reshape(node.shape[0], -1)
    h = act(_lin(P, "node_mlp.0", h))
    h = act(_lin(P, "node_mlp.2", h))
    h = act(_lin(P, "node_mlp.4", h))
    return _lin(P, "node_mlp.6", h)


# --------------------------------------------------------------------------
# losses (src/model/loss.py)
# --------------------------------------------------------------------------
def gradient_penalty(disc: Callable, real_node, real_edge, fake_node, fake_edge, eps_edge, eps_node):
    """loss.py:4-49 with the two torch.rand draws (lines 21-22) passed in."""
    int_node = (eps_node * real_node + (1 - eps_node) * fake_node).requires_grad_(True)
    int_edge = (eps_edge * real_edge + (1 - eps_edge) * fake_edge).requires_grad_(True)
    logits = disc(int_edge, int_node)
    g_node, g_edge = torch.autograd.grad(
        logits, [int_node, int_edge], grad_outputs=torch.ones_like(logits),
        create_graph=True, retain_graph=True)
    b = logits.shape[0]
    flat = torch.cat([g_node.reshape(b, -1), g_edge.reshape(b, -1)], dim=1)
    return ((flat.norm(2, dim=1) - 1) ** 2).mean()


def discriminator_loss(gen: Callable, disc: Callable, drug_adj, drug_annot, mol_adj, mol_annot,
                       lambda_gp: float, eps_edge, eps_node):
    """loss.py:52-72 -> (node, edge, d_loss).  Argument order: adjacency first."""
    real = -disc(drug_adj, drug_annot).mean()
    node, edge, node_sample, edge_sample = gen(mol_adj, mol_annot)
    fake = disc(edge_sample.detach(), node_sample.detach()).mean()
    gp = gradient_penalty(disc, drug_annot, drug_adj, node_sample.detach(), edge_sample.detach(),
                          eps_edge, eps_node)
    return node, edge, fake + real + lambda_gp * gp


def generator_loss(gen: Callable, disc: Callable, mol_adj, mol_annot):
    """loss.py:75-84 -> (g_loss, node, edge, node_sample, edge_sample); raw logits go to D."""
    node, edge, node_sample, edge_sample = gen(mol_adj, mol_annot)
    return -disc(edge_sample, node_sample).mean(), node, edge, node_sample, edge_sample


# --------------------------------------------------------------------------
# nn.Module shells + the GAN step (train.py:351-384)
# --------------------------------------------------------------------------
class OracleNet(torch.nn.Module):
    """nn.Module view of a flat parameter mapping so torch.optim / DDP-style
    code can drive the oracle.  ``kind`` is "G" or "D"."""

    def __init__(self, kind: str, cfg: NetConfig, values: Mapping[str, torch.Tensor]):
        super().__init__()
        self.kind, self.cfg = kind, cfg
        schema = generator_schema(cfg) if kind == "G" else discriminator_schema(cfg)
        self.names = [n for n, _ in schema]
        self.flat = torch.nn.ParameterList(
            [torch.nn.Parameter(torch.as_tensor(values[n]).detach().clone()) for n in self.names])
        for (n, shape), p in zip(schema, self.flat):
            if tuple(p.shape) != tuple(shape):
                raise ValueError(f"{n}: expected {shape}, got {tuple(p.shape)}")

    def named(self) -> Dict[str, torch.Tensor]:
        return dict(zip(self.names, self.flat))

    def forward(self, z_e, z_n):
        fn = generator_forward if self.kind == "G" else discriminator_forward
        return fn(self.named(), z_e, z_n, self.cfg, self.training)


def make_optimizers(G: torch.nn.Module, D: torch.nn.Module, lr: float = 1e-5):
    """train.py:213-214: AdamW(lr, betas=(0.9, 0.999)), default weight decay."""
    return (torch.optim.AdamW(G.parameters(), lr, (0.9, 0.999)),
            torch.optim.AdamW(D.parameters(), lr, (0.9, 0.999)))


def gan_step(G, D, g_opt, d_opt, disc_edge, disc_node, gen_edge, gen_node, lambda_gp, eps_edge, eps_node):
    """One iteration of train.py:351-384 without logging.  Returns (d_loss, g_loss)
    as 0-dim tensors."""
    g_opt.zero_grad(set_to_none=True)
    d_opt.zero_grad(set_to_none=True)
    _, _, d_loss = discriminator_loss(G, D, disc_edge, disc_node, gen_edge, gen_node, lambda_gp,
                                      eps_edge, eps_node)
    d_loss.backward()
    d_opt.step()
    g_opt.zero_grad(set_to_none=True)
    d_opt.zero_grad(set_to_none=True)
    g_loss = generator_loss(G, D, gen_edge, gen_node)[0]
    g_loss.backward()
    g_opt.step()
    return d_loss.detach(), g_loss.detach()
