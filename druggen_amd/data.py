"""Batch densify on the GPU -- the step that feeds the hot path.

Mirror of the reference's ``load_molecules`` / ``label2onehot``
(``src/data/utils.py:15-23,128-142``): COO edges + bond labels of a padded PyG
batch become ``a_tensor [B,N,N,b_dim]`` one-hot float32; node features are
reshaped to ``x_tensor [B,N,m_dim]``.  PyG itself is not needed: any object with
``edge_index [2,E]``, ``edge_attr [E]``, ``x [B*N, m_dim]`` and ``batch [B*N]``
attributes works (``torch_geometric.data.Batch`` has exactly these).
"""
from __future__ import annotations

import torch

from . import _lib
from .functional import _dev, attach_one_hot_labels

__all__ = ["dense_one_hot_adjacency", "load_molecules", "label2onehot", "raise_deferred_checks"]


_pending_checks = []      # (pinned host counter, b_dim, event on the side stream) of batches densified with check="deferred"
_side_streams = {}        # device index -> the stream that carries the counters' device -> host copies


_host_slots = None        # pinned int32 ring for the counters (allocated once: a pinned allocation synchronises the device)
_next_slot = 0


def _side_stream(dev):
    st = _side_streams.get(dev.index)
    if st is None:
        st = _side_streams[dev.index] = torch.cuda.Stream(dev)
    return st


def _host_counter():
    """One pinned int32 of the ring (256 slots; a slot still waiting for its copy after 255 later batches is waited for)."""
    global _host_slots, _next_slot
    if _host_slots is None:
        _host_slots = torch.zeros(256, dtype=torch.int32, pin_memory=True)
    i = _next_slot
    _next_slot = (i + 1) % 256
    for host, _, ev in _pending_checks:
        if host.data_ptr() == _host_slots[i:i + 1].data_ptr():
            ev.synchronize()
    return _host_slots[i:i + 1]


def raise_deferred_checks(wait: bool = False) -> None:
    """Raise for batches densified with ``check="deferred"`` whose bond labels were out of range.  The counters were copied to
    pinned host memory on a side stream behind their kernel; only copies that have already FINISHED are looked at
    (``wait=True``: all of them, after waiting for their events) -- the compute stream is never touched, nothing is enqueued
    and nothing is synchronised."""
    keep, bad_batches = [], []
    for host, b_dim, ev in _pending_checks:
        if wait:
            ev.synchronize()
        elif not ev.query():
            keep.append((host, b_dim, ev))
            continue
        n_bad = int(host[0])      # a plain host read: the copy behind `ev` has completed
        if n_bad:
            bad_batches.append((n_bad, b_dim))
    _pending_checks[:] = keep      # (finished counters are dropped, pending ones stay -- also when this call raises)
    if bad_batches:
        n_bad, b_dim = bad_batches[0]
        more = f" (and {len(bad_batches) - 1} more such batches)" if len(bad_batches) > 1 else ""
        raise RuntimeError(f"{n_bad} adjacency entries of an EARLIER batch had a bond label outside [0, {b_dim}){more} "
                           f"(densified with check='deferred': those labels were clamped into [0, {b_dim - 1}] for the embedding)")


def dense_one_hot_adjacency(edge_index, edge_attr, batch_size: int, vertexes: int, b_dim: int, check=True):
    """``label2onehot(to_dense_adj(edge_index, batch, edge_attr, max_num_nodes=N), b_dim)`` for a
    batch whose graphs are all padded to ``vertexes`` nodes (utils.py:130-137).

    ``check=True`` (default) raises on bond labels outside ``[0, b_dim)`` like the reference's
    ``scatter_`` does (one 4-byte device->host read per batch; the reference's loader syncs per
    batch anyway) and attaches the int32 labels to the result, so that Generator / Discriminator take the
    table-gather edge embedding without re-validating the tensor.  Pass ``check=False`` on a path that must
    not synchronise (the result then carries no labels: it is not known to be one-hot), or ``check="deferred"`` in a
    training loop: no host synchronisation at all -- the labels are clamped into range on the device (so the table gather
    stays memory-safe), attached, and the counter of out-of-range labels is read at a LATER call, once its kernel has
    finished (``raise_deferred_checks``): an invalid batch raises one or two batches late instead of stalling every batch.
    An edge whose endpoints lie in different graphs lands in the SOURCE graph's matrix at column
    ``dst mod N`` -- what ``to_dense_adj`` does with ``dst - ptr[batch[dst]]``."""
    if not edge_index.is_cuda:
        raise RuntimeError("druggen_amd.data runs on the GPU (no CPU fallback)")
    lib = _lib.load()
    dev = edge_index.device
    src = edge_index[0].contiguous().long()
    dst = edge_index[1].contiguous().long()
    attr = edge_attr.reshape(-1).contiguous().long()
    labels = torch.empty(batch_size, vertexes, vertexes, dtype=torch.int32, device=dev)
    a = torch.empty(batch_size, vertexes, vertexes, b_dim, dtype=torch.float32, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    with _dev(a):
        _lib.check(lib.dg_densify(src.data_ptr(), dst.data_ptr(), attr.data_ptr(), src.numel(), batch_size, vertexes,
                                  b_dim, labels.data_ptr(), a.data_ptr(), bad.data_ptr(), _lib.stream_of(a)),
                   "dg_densify")
    if check == "deferred":
        raise_deferred_checks()      # counters of earlier batches whose host copies are complete by now
        # counter -> pinned host memory on a SIDE stream that waits for the densify kernel: the compute stream sees one event
        # record, and the host reads the value only after the copy's own event has completed
        host = _host_counter()
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        side = _side_stream(dev)
        side.wait_event(done)
        with torch.cuda.stream(side):
            host.copy_(bad, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        bad.record_stream(side)
        _pending_checks.append((host, b_dim, ev))
        labels.clamp_(0, b_dim - 1)
        attach_one_hot_labels(a, labels)
    elif check:
        n_bad = int(bad.item())
        if n_bad:
            raise RuntimeError(f"{n_bad} adjacency entries have a bond label outside [0, {b_dim})")
        # one-hot by construction: hand the labels to the model (table-gather edge embedding) without the
        # per-tensor validation sync of ``functional.as_one_hot``
        attach_one_hot_labels(a, labels)
    return a


def label2onehot(labels, dim, device=None):
    """Reference utils.py:15-23 for label tensors already on the GPU."""
    out = torch.zeros(list(labels.size()) + [dim], device=labels.device if device is None else device)
    out.scatter_(len(out.size()) - 1, labels.unsqueeze(-1), 1.)
    return out.float()


def load_molecules(data=None, b_dim=32, m_dim=32, device=None, batch_size=32, check=True):
    """Reference utils.py:128-142 -> (real_graphs, a_tensor, x_tensor).  ``check``: see ``dense_one_hot_adjacency``
    ("deferred" = no host synchronisation per batch)."""
    data = data.to(device) if hasattr(data, "to") and device is not None else data
    vertexes = int(data.batch.shape[0] / batch_size)
    a_tensor = dense_one_hot_adjacency(data.edge_index, data.edge_attr, batch_size, vertexes, b_dim, check=check)
    x_tensor = data.x.view(batch_size, vertexes, -1)
    real_graphs = torch.concat((x_tensor.reshape(batch_size, -1), a_tensor.reshape(batch_size, -1)), dim=-1)
    return real_graphs, a_tensor, x_tensor
