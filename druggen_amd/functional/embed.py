"""Edge embedding MLP + symmetrisation (reference src/model/models.py:57-61,92-94 / 159-163,197-199), its one-hot form (an E-row
table) and the output slots that let several producers write one tensor."""
from __future__ import annotations

import contextlib
import ctypes
import threading
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from ..options import options
from ._runtime import *      # noqa: F401,F403
from .layernorm import *      # noqa: F401,F403
from .dense import *      # noqa: F401,F403
from .heads import *      # noqa: F401,F403
from .ffn import *      # noqa: F401,F403
from .attention import *      # noqa: F401,F403


# --------------------------------------------------------------------------
# edge embedding MLP + symmetrisation (reference models.py:57-61,92-94 / 159-163,197-199)
# --------------------------------------------------------------------------
_ACT_IDS = {"relu": 0, "leaky": 1, "sigmoid": 2, "tanh": 3}
_ACT_FNS = {"relu": torch.relu, "leaky": lambda t: torch.nn.functional.leaky_relu(t, 0.01),
            "sigmoid": torch.sigmoid, "tanh": torch.tanh}


def _embed_packed_w2(w2, dgrad: bool = False):
    key = (id(w2), dgrad)
    hit = _embed_pack_cache.get(key)
    if (hit is not None and hit[0]() is w2 and hit[1] == w2._version and hit[3] == w2.data_ptr()
            and hit[4] == _weights_epoch[0]):
        return hit[2]
    lib = _lib.load()
    n_floats = lib.dg_embed_sym_dgrad_packed_floats() if dgrad else lib.dg_embed_sym_packed_floats()
    packed = torch.empty(int(n_floats), dtype=torch.float32, device=w2.device)
    wd = _c(w2.detach())
    with _dev(w2):
        pack = lib.dg_embed_sym_pack_dgrad if dgrad else lib.dg_embed_sym_pack
        _lib.check(pack(_lib.ptr(wd), _lib.ptr(packed), _lib.stream_of(w2)), "dg_embed_sym_pack")
    _embed_pack_cache[key] = (weakref.ref(w2), w2._version, packed, w2.data_ptr(), _weights_epoch[0])
    return packed


def _composite_embed_sym(a, w1, b1, w2, b2, act):
    f = _ACT_FNS[act]
    h = f(linear(a, w1, b1))
    e = f(linear(h, w2, b2))
    return (e + e.permute(0, 2, 1, 3)) / 2


class OutSlot:
    """A destination buffer handed to an embedding Function as a plain Python object (autograd never sees it): the
    kernel writes into ``tensor`` and the Function returns a fresh alias of it.  Used to let the parts of a batch (a
    one-hot real half, a dense generated half) land in ONE [sum B, N, N, C] buffer without a concatenation copy."""
    __slots__ = ("tensor",)

    def __init__(self, tensor):
        self.tensor = tensor


def _take_slot(slot, shape, dtype, device):
    if slot is None:
        return torch.empty(*shape, dtype=dtype, device=device)
    t = slot.tensor
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device or not t.is_contiguous():
        raise RuntimeError("OutSlot does not match the embedding's output")
    return t.view(shape)          # a fresh alias (forward runs with grad mode off)


class _JoinParts(Function):
    """The buffer whose dim-0 slices were filled by ``parts`` as ONE tensor of the graph: forward returns an alias of the
    buffer (no copy), backward hands each part its slice of the gradient (views)."""

    @staticmethod
    def forward(ctx, slot, *parts):
        ctx.sizes = [p.shape[0] for p in parts]
        off = 0
        for p_ in parts:
            if p_.data_ptr() != slot.tensor[off:off + p_.shape[0]].data_ptr():
                raise RuntimeError("join_parts: a part does not live in its slice of the buffer")
            off += p_.shape[0]
        return slot.tensor.view(slot.tensor.shape)

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for n in ctx.sizes:
            outs.append(g[off:off + n])
            off += n
        return (None, *outs)


def join_parts(slot, parts):
    return _JoinParts.apply(slot, *parts)


class _EmbedSym(Function):
    @staticmethod
    def forward(ctx, a, w1, b1, w2, b2, act, out_dtype, slot=None):
        a = _c(a)
        B, N, _, E = a.shape
        H, C = w1.shape[0], w2.shape[0]
        lib = _lib.load()
        out = _take_slot(slot, (B, N, N, C), out_dtype, a.device)
        with _dev(a):
            _lib.check(lib.dg_embed_sym_fwd(_lib.fptr(a), _lib.fptr(_c(w1)), _lib.fptr(_c(b1)), _lib.fptr(_embed_packed_w2(w2)),
                                            _lib.fptr(_c(b2)), _lib.ptr(out), B, N, E, H, C, _ACT_IDS[act],
                                            _lib.dt(out), _lib.stream_of(a)), "dg_embed_sym_fwd")
        _account("embed_sym", B * N * N * (4 * E + out.element_size() * C), 2 * B * N * N * (E * H + H * C))
        ctx.save_for_backward(a, w1, b1, w2, b2)
        ctx.act = act
        ctx.out_dtype = out_dtype
        return out

    @staticmethod
    def backward(ctx, g):
        a, w1, b1, w2, b2 = ctx.saved_tensors
        act = ctx.act
        if torch.is_grad_enabled():
            odt = ctx.out_dtype
            if act in _PIECEWISE_LINEAR:       # native second order (gradient penalty)
                outs = _EmbedSymBwd.apply(a, w1, b1, w2, b2, g, act, odt, ctx.needs_input_grad[0],
                                          ctx.needs_input_grad[1] and not _inputs_only())
                return tuple(outs) + (None, None, None)
            return _double_backward_fallback(lambda *t: _composite_embed_sym(*t, act).to(odt), (a, w1, b1, w2, b2), g) + (None, None, None)
        return _embed_bwd_launch(a, w1, b1, w2, b2, g, act, ctx.out_dtype, ctx.needs_input_grad[0],
                                 ctx.needs_input_grad[1] and not _inputs_only()) + (None, None, None)


_PIECEWISE_LINEAR = ("relu", "leaky")


def _embed_bwd_launch(a, w1, b1, w2, b2, g, act, out_dtype, need_da, need_w):
    B, N, _, E = a.shape
    H, C = w1.shape[0], w2.shape[0]
    lib = _lib.load()
    g = _c(g if g.dtype == out_dtype else g.to(out_dtype))
    da = torch.empty_like(a) if need_da else None
    dw1, db1, dw2, db2 = (torch.empty_like(t) for t in (w1, b1, w2, b2))
    if (out_dtype == torch.bfloat16 and act in _PIECEWISE_LINEAR and E <= 8 and N <= 48
            and options.embed_bf16 == "fast"):
        # bf16 gradients, relu / leaky: row-block streaming kernel (csrc/embed_bf16.hip)
        need = int(lib.dg_embed_sym_bwd_bf16_workspace_bytes(B, N))
        with _dev(a):
            ws = _scratch(a, need, "embed16")
            _lib.check(lib.dg_embed_sym_bwd_bf16(_lib.fptr(a), _lib.fptr(_c(w1)), _lib.fptr(_c(b1)), _lib.fptr(_c(w2)),
                                                 _lib.fptr(_c(b2)), _lib.ptr(g), _lib.ptr(da), _lib.ptr(dw1), _lib.ptr(db1),
                                                 _lib.ptr(dw2), _lib.ptr(db2), ws.data_ptr(), ws.numel(), B, N, E, H, C,
                                                 _ACT_IDS[act], _lib.stream_of(a)), "dg_embed_sym_bwd_bf16")
        _account("embed_sym", B * N * N * (4 * E * (2 if da is not None else 1) + 2 * g.element_size() * C),
                 2 * B * N * N * (E * H + H * C) * 3)
        if not need_w:
            dw1 = db1 = dw2 = db2 = None
        return da, dw1, db1, dw2, db2
    need = int(lib.dg_embed_sym_workspace_bytes(B, N))
    with _dev(a):
        ws = _scratch(a, need, "embed")
        _lib.check(lib.dg_embed_sym_bwd(_lib.fptr(a), _lib.fptr(_c(w1)), _lib.fptr(_c(b1)),
                                        _lib.fptr(_embed_packed_w2(w2)), _lib.fptr(_embed_packed_w2(w2, True)),
                                        _lib.fptr(_c(b2)), _lib.ptr(g), _lib.ptr(da), _lib.ptr(dw1), _lib.ptr(db1),
                                        _lib.ptr(dw2), _lib.ptr(db2), ws.data_ptr(), ws.numel(), B, N, E, H, C,
                                        _ACT_IDS[act], _lib.dt(g), _lib.stream_of(a)), "dg_embed_sym_bwd")
    _account("embed_sym", B * N * N * (4 * E * (2 if da is not None else 1) + g.element_size() * C),
             2 * B * N * N * (E * H + H * C) * 3)
    if not need_w:
        dw1 = db1 = dw2 = db2 = None
    return da, dw1, db1, dw2, db2


class _EmbedSymBwd(Function):
    """First backward of ``_EmbedSym`` as a differentiable node (piecewise-linear activations): its own backward is
    ``dg_embed_sym_bwd2`` -- the gradient penalty differentiates d out / d a (reference loss.py:32-39).  Only the
    adjoint of ``da`` is propagated; adjoints of the parameter gradients would need the composite graph."""

    @staticmethod
    def forward(ctx, a, w1, b1, w2, b2, g, act, out_dtype, need_da, need_w):
        ctx.save_for_backward(a, w1, b1, w2, b2, g)
        ctx.act, ctx.out_dtype = act, out_dtype
        outs = _embed_bwd_launch(a.detach(), w1.detach(), b1.detach(), w2.detach(), b2.detach(), g.detach(), act,
                                 out_dtype, need_da, need_w)
        ctx.mark_non_differentiable(*[o for o in outs[1:] if o is not None])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, t_da, t_dw1, t_db1, t_dw2, t_db2):
        a, w1, b1, w2, b2, g = ctx.saved_tensors
        if t_da is None:
            return (None,) * 10
        B, N, _, E = a.shape
        H, C = w1.shape[0], w2.shape[0]
        lib = _lib.load()
        g = _c(g if g.dtype == ctx.out_dtype else g.to(ctx.out_dtype))
        t = _c(t_da.float())
        gg = torch.empty_like(g)
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        need = int(lib.dg_embed_sym_workspace_bytes(B, N))
        with _dev(a):
            ws = _scratch(a, need, "embed")
            _lib.check(lib.dg_embed_sym_bwd2(_lib.fptr(_c(a)), _lib.fptr(_c(w1)), _lib.fptr(_c(b1)),
                                             _lib.fptr(_embed_packed_w2(w2)), _lib.fptr(_embed_packed_w2(w2, True)),
                                             _lib.fptr(_c(b2)), _lib.ptr(g), _lib.fptr(t), _lib.ptr(gg), _lib.ptr(gw1),
                                             _lib.ptr(gw2), ws.data_ptr(), ws.numel(), B, N, E, H, C,
                                             _ACT_IDS[ctx.act], _lib.dt(g), _lib.stream_of(a)), "dg_embed_sym_bwd2")
        _account("embed_sym", B * N * N * (8 * E + 2 * g.element_size() * C), 2 * B * N * N * (E * H + H * C) * 4)
        if _inputs_only() or not ctx.needs_input_grad[1]:
            gw1 = gw2 = None
        return None, gw1, None, gw2, None, gg, None, None, None, None


def embed_sym(a, w1, b1, w2, b2, act: str, out_dtype=torch.float32, slot=None):
    """(f(a) + f(a)^T(i<->j)) / 2 with f = act(W2 act(W1 a + b1) + b2): the edge embedding MLP and the
    symmetrisation of Generator / Discriminator in one kernel per direction (hidden 64, dim 128).
    The input graph ``a`` is float32; the [B,N,N,dim] result is stored as ``out_dtype``."""
    ok = (a.is_cuda and a.dtype == torch.float32 and a.dim() == 4 and a.shape[1] == a.shape[2] and act in _ACT_IDS
          and a.shape[-1] <= 16 and tuple(w1.shape) == (64, a.shape[-1]) and tuple(w2.shape) == (128, 64)
          and b1 is not None and b2 is not None)
    if not ok or (in_second_order_forward() and act not in _PIECEWISE_LINEAR):
        out = _composite_embed_sym(a, w1, b1, w2, b2, act).to(out_dtype)
        if slot is not None:
            slot.tensor = None          # the caller falls back to a concatenation
        return out
    return _EmbedSym.apply(a, w1, b1, w2, b2, act, out_dtype, slot)


# --------------------------------------------------------------------------
# one-hot input graphs: the embedding MLP collapses to an E-row table
# (reference src/data/utils.py:15-23 + models.py:57-61,92-94)
# --------------------------------------------------------------------------
def as_one_hot(a, labels=None):
    """Declare (after checking it) that the edge tensor ``a`` [B,N,N,E] is one-hot over its last dim -- true for
    every adjacency the reference's ``load_molecules`` / ``label2onehot`` produces (generator input, the
    discriminator's real batch), false for generated / interpolated tensors.  The int32 labels are attached to the
    tensor object; Generator / Discriminator then evaluate the edge-embedding MLP on the E distinct rows only
    (``dg_onehot_embed_fwd/bwd``).  The check costs one device->host read per NEW tensor object (the result is
    cached on it together with the tensor's version counter: an in-place write invalidates it), so a resident batch
    is checked once.  Returns ``a``."""
    if not (torch.is_tensor(a) and a.is_cuda and a.dim() == 4):
        return a
    if getattr(a, "_dg_labels", None) is not None and getattr(a, "_dg_labels_version", None) == a._version:
        return a                      # same object, not written since the check (in-place updates bump _version)
    a._dg_labels_version = a._version
    if a.requires_grad or a.dtype != torch.float32:
        a._dg_labels = False
        return a
    with torch.no_grad():
        if labels is None:
            labels = a.argmax(-1).to(torch.int32)
        ok = ((a.amax(-1) == 1) & (a.sum(-1) == 1) & (a.amin(-1) == 0)).all() if a.shape[-1] > 1 else (a == 1).all()
    a._dg_labels = labels.contiguous() if bool(ok) else False     # the only host sync: once per tensor object
    return a


def attach_one_hot_labels(a, labels):
    """Declare WITHOUT checking that ``a`` [B,N,N,E] is one-hot with the given int32 ``labels`` [B,N,N] -- for producers
    that build ``a`` from the labels (``data.dense_one_hot_adjacency``: reference utils.py:15-23,130-137) or refresh both
    together (``GraphedGANStep``).  No device->host sync.  Returns ``a``."""
    if labels.dtype != torch.int32 or tuple(labels.shape) != tuple(a.shape[:-1]) or labels.device != a.device:
        raise ValueError("labels must be an int32 tensor on a's device with a's shape minus the last dim")
    a._dg_labels = labels if labels.is_contiguous() else labels.contiguous()
    a._dg_labels_version = a._version
    return a


def one_hot_labels(a):
    """The int32 labels attached by ``as_one_hot`` (None for tensors that are not declared one-hot)."""
    lab = getattr(a, "_dg_labels", None)
    if not torch.is_tensor(lab) or getattr(a, "_dg_labels_version", None) != a._version:
        return None                   # never declared, not one-hot, or written in place since the check
    return lab


class _OneHotEmbed(Function):
    @staticmethod
    def forward(ctx, labels, table, out_dtype, slot=None):
        B, N = labels.shape[0], labels.shape[1]
        E, C = table.shape
        lib = _lib.load()
        table = _c(table)
        out = _take_slot(slot, (B, N, N, C), out_dtype, labels.device)
        with _dev(labels):
            _lib.check(lib.dg_onehot_embed_fwd(labels.data_ptr(), _lib.fptr(table), _lib.ptr(out), B, N, E, C, _lib.dt(out),
                                               _lib.stream_of(out)), "dg_onehot_embed_fwd")
        _account("embed_sym", B * N * N * (8 + out.element_size() * C))
        ctx.save_for_backward(labels)
        ctx.shape = (E, C, out_dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (labels,) = ctx.saved_tensors
        E, C, out_dtype = ctx.shape
        B, N = labels.shape[0], labels.shape[1]
        lib = _lib.load()
        g = _c(g if g.dtype == out_dtype else g.to(out_dtype))
        dtable = torch.empty(E, C, dtype=torch.float32, device=g.device)
        need = int(lib.dg_onehot_embed_workspace_bytes(E, C))
        with _dev(g):
            ws = _scratch(g, need, "onehot")
            _lib.check(lib.dg_onehot_embed_bwd(labels.data_ptr(), _lib.ptr(g), _lib.ptr(dtable), ws.data_ptr(), ws.numel(),
                                               B, N, E, C, _lib.dt(g), _lib.stream_of(g)), "dg_onehot_embed_bwd")
        _account("embed_sym", B * N * N * (8 + g.element_size() * C))
        return None, dtable, None, None


def embed_sym_onehot(labels, w1, b1, w2, b2, act: str, out_dtype=torch.float32, slot=None):
    """``embed_sym`` for a one-hot input given by its labels [B,N,N]: the MLP runs on the E unit vectors (plain torch
    ops on [E,64] / [E,128] tensors, differentiated by autograd), the [B,N,N,dim] result is a symmetrised gather."""
    f = _ACT_FNS[act]
    table = f(torch.nn.functional.linear(f(w1.t() + b1), w2, b2))      # [E, dim]: row c = f(one_hot(c))
    return _OneHotEmbed.apply(labels, table, out_dtype, slot)


__all__ = [_n for _n in dir() if not _n.startswith("__")]
