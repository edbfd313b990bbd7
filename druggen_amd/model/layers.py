"""MI355X-native mirror of the reference's ``src/model/layers.py``.

Module names, parameter names and call signatures follow the reference so its
checkpoints load unchanged; the compute goes through the HIP kernels of
``libdruggen_hip.so`` (``druggen_amd.functional``): every dim-128 / 384 ``nn.Linear``
runs on ``dg_row_gemm`` / ``dg_linear_wgrad`` (fp16 hi + lo split MFMA, fp32 class); only odd
shapes (the Discriminator head, tiny test models) reach the ROCm BLAS.  There is no CPU path.
"""
from __future__ import annotations

import math

import torch.nn as nn
from torch.nn import functional as F

from .. import functional as dgf


class MLP(nn.Module):
    """fc2(ReLU(fc1(x))) then dropout -- reference layers.py:7-54 (always ReLU)."""

    def __init__(self, in_feat, hid_feat=None, out_feat=None, dropout=0.):
        super().__init__()
        hid_feat = hid_feat or in_feat
        out_feat = out_feat or in_feat
        self.fc1 = nn.Linear(in_feat, hid_feat)
        self.act = nn.ReLU()
        self.fc2 = nn.Linear(hid_feat, out_feat)
        self.droprateout = nn.Dropout(dropout)

    def forward(self, x):
        h = dgf.linear_relu(x, self.fc1.weight, self.fc1.bias)
        return self.droprateout(dgf.linear(h, self.fc2.weight, self.fc2.bias))

    def ffn_ln_args(self, ln):
        return (self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, ln.weight, ln.bias, ln.eps)

    def forward_residual_ln(self, x, ln, want_handle=False):
        """ln(x + self(x)) -- Encoder_Block lines 191-192 of the reference -- with fc2, the
        residual add and the LayerNorm in one kernel when dropout is inactive.  ``want_handle``: also returns
        the ``dgf.LNHandle`` of that LayerNorm (or None) for the next block's attention node."""
        if self.droprateout.p > 0.0 and self.training:
            y = dgf.ln_residual(x, self.forward(x), ln.weight, ln.bias, ln.eps)
            return (y, None) if want_handle else y
        return dgf.ffn_ln(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                          ln.weight, ln.bias, ln.eps, want_handle=want_handle)


class MHA(nn.Module):
    """Edge-modulated per-channel attention -- reference layers.py:56-137.

    Parameters q, k, v, e, out_e, out_n (six ``Linear(dim, dim)``).  The score
    s = q_i k_j / sqrt(d_k) * (e^2 + e), its softmax over j and the aggregation
    with v run in one fused HIP kernel (``dg_attn_core_fwd``).
    """

    def __init__(self, dim, heads, attention_dropout=0.):
        super().__init__()
        assert dim % heads == 0
        self.heads = heads
        self.scale = 1. / math.sqrt(dim)      # kept for parity; unused (reference layers.py:84)
        self.q = nn.Linear(dim, dim)
        self.k = nn.Linear(dim, dim)
        self.v = nn.Linear(dim, dim)
        self.e = nn.Linear(dim, dim)
        self.d_k = dim // heads
        self.out_e = nn.Linear(dim, dim)
        self.out_n = nn.Linear(dim, dim)

    def forward(self, node, edge, need_edge=True, raw=False):
        """``raw=True`` returns the attention aggregates (o, s) before out_n / out_e so the
        caller can fuse those projections with its residual + LayerNorm."""
        lin = dgf.linear
        q = lin(node, self.q.weight, self.q.bias)
        k = lin(node, self.k.weight, self.k.bias)
        v = lin(node, self.v.weight, self.v.bias)
        e = lin(edge, self.e.weight, self.e.bias)
        s, o = dgf.attn_core(q, k, v, e, 1.0 / math.sqrt(self.d_k), need_s=need_edge)
        if raw:
            return o, s
        node_out = lin(o, self.out_n.weight, self.out_n.bias)
        edge_out = lin(s, self.out_e.weight, self.out_e.bias) if need_edge else None
        return node_out, edge_out


class Encoder_Block(nn.Module):
    """Reference layers.py:139-193.  LayerNorms ln1, ln3, ln4, ln5, ln6 (no ln2);
    the node residual uses the normalised input x1."""

    def __init__(self, dim, heads, act, mlp_ratio=4, drop_rate=0.):
        super().__init__()
        self.ln1 = nn.LayerNorm(dim)
        self.attn = MHA(dim, heads, drop_rate)
        self.ln3 = nn.LayerNorm(dim)
        self.ln4 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, dim * mlp_ratio, dim, dropout=drop_rate)
        self.mlp2 = MLP(dim, dim * mlp_ratio, dim, dropout=drop_rate)
        self.ln5 = nn.LayerNorm(dim)
        self.ln6 = nn.LayerNorm(dim)

    @staticmethod
    def _ln(ln, a, r=None):
        return dgf.ln_residual(a, r, ln.weight, ln.bias, ln.eps)

    def forward(self, x, y, need_edge=True, y_ln=None, want_ln=False):
        """``y_ln`` / ``want_ln`` (TransformerEncoder): the handle of the LayerNorm that produced ``y`` goes in, the
        handle of ln6 comes out as a third result -- the next block runs ln6's backward inside its own dy GEMM."""
        x1 = self._ln(self.ln1, x)
        # q/k/v/e projections, attention core, out_n/out_e + residual + ln3/ln4: one autograd node
        x2, y2 = dgf.attn_block(x1, y, self.attn, self.ln3, self.ln4, need_edge, y_ln=y_ln)
        if not need_edge:
            x = self.mlp.forward_residual_ln(x2, self.ln5)
            return (x, None, None) if want_ln else (x, None)
        if not (self.training and (self.mlp.droprateout.p > 0.0 or self.mlp2.droprateout.p > 0.0)):
            # both feed-forward halves as one node: the node-level launches ride in the edge-level ones
            x, y, handle = dgf.ffn_ln_pair(x2, self.mlp.ffn_ln_args(self.ln5), y2, self.mlp2.ffn_ln_args(self.ln6))
            return (x, y, handle) if want_ln else (x, y)
        x = self.mlp.forward_residual_ln(x2, self.ln5)
        y, handle = self.mlp2.forward_residual_ln(y2, self.ln6, want_handle=True)
        return (x, y, handle) if want_ln else (x, y)


class TransformerEncoder(nn.Module):
    """Reference layers.py:195-234: ``depth`` blocks in ``Encoder_Blocks``."""

    def __init__(self, dim, depth, heads, act, mlp_ratio=4, drop_rate=0.1):
        super().__init__()
        self.Encoder_Blocks = nn.ModuleList([
            Encoder_Block(dim, heads, act, mlp_ratio, drop_rate) for _ in range(depth)
        ])

    def forward(self, x, y, need_edge=True):
        """``need_edge=False`` skips the edge branch of the LAST block (its output
        is dropped by the Discriminator, reference models.py:202-207)."""
        last = len(self.Encoder_Blocks) - 1
        y_ln = None
        for idx, block in enumerate(self.Encoder_Blocks):
            x, y, y_ln = block(x, y, need_edge or idx != last, y_ln=y_ln, want_ln=True)
        return x, y
