"""ORACLE (test infrastructure only) for the steps either side of the hot path.

numpy restatements of
  * PyG 2.2.0 ``torch_geometric.utils.to_dense_adj`` (pinned in the reference's
    ``environment.yml:171``; the package is absent here, so its published
    algorithm is restated: index = (batch[src], src - ptr[batch[src]],
    dst - ptr[batch[dst]]), values scatter-ADDed) followed by the reference's
    ``label2onehot`` (``src/data/utils.py:15-23``) as called from
    ``load_molecules`` (``src/data/utils.py:128-142``).  The reference owns no vectors
    for it; pinned to PyG's own published ones -- the "Examples" of the
    ``to_dense_adj`` docstring -- in ``tests/test_host.py``;
  * ``torch.optim.AdamW`` single-tensor update (reference ``train.py:213-214``);
  * ``torch.max(x, -1)[1]`` (reference ``inference.py:197-198``).
"""
from __future__ import annotations

import numpy as np


def to_dense_adj(edge_index, batch, edge_attr, max_num_nodes):
    edge_index, batch, edge_attr = (np.asarray(v) for v in (edge_index, batch, edge_attr))
    B = int(batch.max()) + 1 if batch.size else 0
    num_nodes = np.bincount(batch, minlength=B)
    ptr = np.concatenate([[0], np.cumsum(num_nodes)])
    b = batch[edge_index[0]]
    i = edge_index[0] - ptr[b]
    j = edge_index[1] - ptr[batch[edge_index[1]]]
    adj = np.zeros((B, max_num_nodes, max_num_nodes), dtype=np.int64)
    np.add.at(adj, (b, i, j), edge_attr)
    return adj


def label2onehot(labels, dim):
    out = np.zeros(labels.shape + (dim,), dtype=np.float32)
    np.put_along_axis(out, labels[..., None], 1.0, axis=-1)
    return out


def load_molecules(edge_index, edge_attr, x, batch, batch_size, b_dim):
    n = batch.shape[0] // batch_size
    a = label2onehot(to_dense_adj(edge_index, batch, edge_attr, n), b_dim)
    x_t = np.asarray(x, dtype=np.float32).reshape(batch_size, n, -1)
    graphs = np.concatenate([x_t.reshape(batch_size, -1), a.reshape(batch_size, -1)], axis=-1)
    return graphs, a, x_t


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p = p - (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)
    return p, m, v


def argmax_last(x):
    return np.argmax(np.asarray(x), axis=-1)
