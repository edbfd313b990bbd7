"""MI355X-native mirror of the reference's ``src/model/models.py``."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as dgf
from .layers import TransformerEncoder


def _activation(act):
    """String -> module, as the reference does (models.py:39-46)."""
    table = {"relu": nn.ReLU, "leaky": nn.LeakyReLU, "sigmoid": nn.Sigmoid, "tanh": nn.Tanh}
    return table[act]() if isinstance(act, str) and act in table else act


class _Trunk(nn.Module):
    """Shared front of Generator and Discriminator: node/edge embedding MLPs,
    edge symmetrisation, TransformerEncoder (reference models.py:52-65, 91-97)."""

    def __init__(self, act, vertexes, edges, nodes, dropout, dim, depth, heads, mlp_ratio):
        super().__init__()
        self.vertexes, self.edges, self.nodes = vertexes, edges, nodes
        self.depth, self.dim, self.heads = depth, dim, heads
        self.mlp_ratio, self.dropout = mlp_ratio, dropout
        act = _activation(act)
        self.features = vertexes * vertexes * edges + vertexes * nodes
        self.transformer_dim = vertexes * vertexes * dim + vertexes * dim
        self.node_layers = nn.Sequential(nn.Linear(nodes, 64), act, nn.Linear(64, dim), act, nn.Dropout(dropout))
        self.edge_layers = nn.Sequential(nn.Linear(edges, 64), act, nn.Linear(64, dim), act, nn.Dropout(dropout))
        self.TransformerEncoder = TransformerEncoder(dim=dim, depth=depth, heads=heads, act=act,
                                                     mlp_ratio=mlp_ratio, drop_rate=dropout)
        self._act = act
        self._act_name = {nn.ReLU: "relu", nn.LeakyReLU: "leaky", nn.Sigmoid: "sigmoid", nn.Tanh: "tanh"}.get(type(act))
        # the fused kernels hard-code nn.LeakyReLU's default slope (what the reference's "leaky" builds, models.py:42): a
        # user-supplied module with another slope takes the composite torch path
        if isinstance(act, nn.LeakyReLU) and float(act.negative_slope) != 0.01:
            self._act_name = None

    def _embed(self, seq, z):
        """Linear(in,64) - act - Linear(64,dim) - act - Dropout (models.py:52-61); the
        second Linear's weight gradient (64 -> dim over all edge rows) uses dg_linear_wgrad."""
        if (not (self.training and self.dropout > 0.0)) and dgf.node_embed_supported(z, seq[0], seq[2], self._act_name):
            return dgf.node_embed(z, seq[0], seq[2], self._act_name)      # both layers: one launch per direction
        h = self._act(dgf.linear(z, seq[0].weight, seq[0].bias))
        h = self._act(dgf.linear(h, seq[2].weight, seq[2].bias))
        return seq[4](h)

    def _embed_edges(self, z_e, adt, slot=None):
        """Linear(E,64) - act - Linear(64,dim) - act - symmetrise (models.py:57-61,92-94).  ``slot``: write the result
        into that buffer (``dgf.OutSlot``); paths that cannot set ``slot.tensor = None``."""
        el = self.edge_layers
        if self._act_name is None or (self.training and self.dropout > 0.0):
            edge = self._embed(el, z_e)
            if slot is not None:
                slot.tensor = None
            return ((edge + edge.permute(0, 2, 1, 3)) / 2).to(adt)
        labels = dgf.one_hot_labels(z_e)
        if (labels is not None and not z_e.requires_grad and not dgf.in_second_order_forward()
                and el[2].weight.shape[0] == 128 and z_e.shape[-1] <= 16):
            # one-hot graph (dataset batch): E distinct embeddings -> table gather (dg_onehot_embed_fwd)
            return dgf.embed_sym_onehot(labels, el[0].weight, el[0].bias, el[2].weight, el[2].bias, self._act_name, adt, slot)
        # one kernel per direction (dg_embed_sym_fwd / _bwd)
        return dgf.embed_sym(z_e, el[0].weight, el[0].bias, el[2].weight, el[2].bias, self._act_name, adt, slot)

    def _encode(self, z_e, z_n, need_edge):
        """``z_e`` may be a tuple of edge tensors that together form the batch (e.g. a one-hot real batch and a
        dense generated one): each part takes its own embedding path, the encoder sees one batch."""
        parts = tuple(z_e) if isinstance(z_e, (tuple, list)) else (z_e,)
        if not parts[0].is_cuda:
            raise RuntimeError("druggen_amd modules run on MI355X only (no CPU fallback): move the model and "
                               "its inputs to a GPU device")
        adt = dgf.activation_dtype()      # storage of the encoder activations (float32, or bfloat16: configs[2])
        node = self._embed(self.node_layers, z_n)
        if node.dtype != adt:
            node = node.to(adt)
        if len(parts) == 1:
            edge = self._embed_edges(parts[0], adt)
        else:
            # every part writes its embedding into its dim-0 slice of ONE buffer: no concatenation copy of [B,N,N,C] tensors
            N, C = parts[0].shape[1], self.edge_layers[2].weight.shape[0]
            total = sum(p.shape[0] for p in parts)
            full = torch.empty(total, N, N, C, dtype=adt, device=parts[0].device)
            slots, edges, off = [], [], 0
            for p in parts:
                slots.append(dgf.OutSlot(full[off:off + p.shape[0]]))
                edges.append(self._embed_edges(p, adt, slots[-1]))
                off += p.shape[0]
            if all(sl.tensor is not None for sl in slots):
                edge = dgf.join_parts(dgf.OutSlot(full), edges)
            else:
                edge = torch.cat(edges)
        return self.TransformerEncoder(node, edge, need_edge)


class Generator(_Trunk):
    """Reference models.py:5-103.  ``forward(z_e, z_n)`` -- edge tensor first --
    returns (node, edge, node_sample, edge_sample) with raw logits."""

    def __init__(self, act, vertexes, edges, nodes, dropout, dim, depth, heads, mlp_ratio):
        super().__init__(act, vertexes, edges, nodes, dropout, dim, depth, heads, mlp_ratio)
        self.readout_e = nn.Linear(dim, edges)
        self.readout_n = nn.Linear(dim, nodes)
        self.softmax = nn.Softmax(dim=-1)     # never applied (reference models.py:69)

    def forward(self, z_e, z_n):
        node, edge = self._encode(z_e, z_n, True)
        # logits are float32 in every activation mode (they are the model's outputs and D's inputs)
        node_sample = dgf.readout(node, self.readout_n.weight, self.readout_n.bias)
        edge_sample = dgf.readout(edge, self.readout_e.weight, self.readout_e.bias)
        return node, edge, node_sample, edge_sample


class Discriminator(_Trunk):
    """Reference models.py:106-209.  ``forward(z_e, z_n)`` -> logits [B, 1]."""

    def __init__(self, act, vertexes, edges, nodes, dropout, dim, depth, heads, mlp_ratio):
        super().__init__(act, vertexes, edges, nodes, dropout, dim, depth, heads, mlp_ratio)
        act = self._act
        self.node_features = vertexes * dim
        self.edge_features = vertexes * vertexes * dim
        self.node_mlp = nn.Sequential(nn.Linear(self.node_features, 64), act, nn.Linear(64, 32), act,
                                      nn.Linear(32, 16), act, nn.Linear(16, 1))

    def forward(self, z_e, z_n):
        node, _ = self._encode(z_e, z_n, False)
        h = node.reshape(node.shape[0], -1).float()
        mlp = self.node_mlp
        tail = (mlp[2], mlp[4], mlp[6])
        z1 = mlp[0](h)
        if dgf.head_tail_supported(z1, tail, self._act_name):
            # act - Linear(64, 32) - act - Linear(32, 16) - act - Linear(16, 1): one launch per direction (dg_head_*)
            return dgf.head_tail(z1, tail, self._act_name)
        return mlp[6](mlp[5](mlp[4](mlp[3](mlp[2](mlp[1](z1))))))


class simple_disc(nn.Module):
    """Reference models.py:212-269: plain MLP critic (imported by train.py:20,
    never instantiated there).  Plain torch, kept importable for drop-in parity."""

    def __init__(self, act, m_dim, vertexes, b_dim):
        super().__init__()
        if act not in ("relu", "leaky", "sigmoid", "tanh"):
            raise ValueError("Unsupported activation function: {}".format(act))
        act = _activation(act)
        widths = [vertexes * m_dim + vertexes * vertexes * b_dim, 256, 128, 64, 32, 16, 1]
        layers = []
        for i in range(6):
            layers.append(nn.Linear(widths[i], widths[i + 1]))
            if i < 5:
                layers.append(act)
        self.predictor = nn.Sequential(*layers)

    def forward(self, x):
        return self.predictor(x)
