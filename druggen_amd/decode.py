"""Argmax decode on the GPU (reference ``inference.py:197-198``,
``src/util/utils.py:220-221``): logits -> compact uint8 label tensors, so the
CPU-side RDKit ``matrices2mol`` receives bytes instead of float logits."""
from __future__ import annotations

import torch

from . import _lib
from .functional import _c, _dev

__all__ = ["argmax_labels", "decode_molecule_labels"]


def argmax_labels(logits):
    """``torch.max(logits, -1)[1]`` as uint8 (classes <= 255)."""
    if not logits.is_cuda:
        raise RuntimeError("druggen_amd.decode runs on the GPU (no CPU fallback)")
    x = _c(logits.detach())
    E = x.shape[-1]
    rows = x.numel() // E
    out = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
    with _dev(x):
        _lib.check(_lib.load().dg_argmax_decode(_lib.ptr(x), rows, E, out.data_ptr(), _lib.stream_of(x)),
                   "dg_argmax_decode")
    return out


def decode_molecule_labels(node_sample, edge_sample):
    """(atom labels [B,N], bond labels [B,N,N]) from the Generator's logits."""
    return argmax_labels(node_sample), argmax_labels(edge_sample)
