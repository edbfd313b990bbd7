"""Graph attention (reference src/model/layers.py:97-137,185-190): the attention core with its two backward orders and the whole
attention half of an Encoder_Block as one autograd node (float32: dg_attn_half_f32_*; bf16: dg_attn_half_*)."""
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


# --------------------------------------------------------------------------
# graph attention core  (reference src/model/layers.py:119-134)
# --------------------------------------------------------------------------
def _attn_shapes(q, e):
    B, N, C = q.shape
    if tuple(e.shape) != (B, N, N, C):
        raise RuntimeError(f"attn_core: edge tensor {tuple(e.shape)} does not match node tensor {tuple(q.shape)}")
    return B, N, C


class _AttnCore(Function):
    @staticmethod
    def forward(ctx, q, k, v, e, alpha, need_s):
        q, k, v, e = _c(q), _c(k), _c(v), _c(e)
        B, N, C = _attn_shapes(q, e)
        lib = _lib.load()
        s = torch.empty_like(e) if need_s else None
        o = torch.empty_like(q)
        with _dev(q):
            _lib.check(lib.dg_attn_core_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(s),
                                            _lib.ptr(o), B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)), "dg_attn_core_fwd")
        _account("attn_fwd", q.element_size() * B * ((2 if need_s else 1) * N * N * C + 4 * N * C))
        ctx.save_for_backward(q, k, v, e)
        ctx.alpha = alpha
        ctx.set_materialize_grads(False)
        if s is None:
            s = q.new_empty(0)
            ctx.mark_non_differentiable(s)
        return s, o

    @staticmethod
    def backward(ctx, ws, wo):
        q, k, v, e = ctx.saved_tensors
        if wo is None:
            wo = torch.zeros_like(q)
        if ws is not None and ws.numel() == 0:
            ws = None
        dq, dk, dv, de = _AttnCoreBwd.apply(q, k, v, e, ws, wo, ctx.alpha)
        return dq, dk, dv, de, None, None


class _AttnCoreBwd(Function):
    @staticmethod
    def forward(ctx, q, k, v, e, ws, wo, alpha):
        B, N, C = _attn_shapes(q, e)
        ws = None if ws is None else _c(ws)
        wo = _c(wo)
        lib = _lib.load()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        de = torch.empty_like(e)
        with _dev(q):
            _lib.check(lib.dg_attn_core_bwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(ws),
                                            _lib.ptr(wo), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(de),
                                            B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)), "dg_attn_core_bwd")
        _account("attn_bwd", q.element_size() * B * ((3 if ws is not None else 2) * N * N * C + 7 * N * C))
        ctx.save_for_backward(q, k, v, e, ws, wo)
        ctx.alpha = alpha
        return dq, dk, dv, de

    @staticmethod
    @once_differentiable
    def backward(ctx, tq, tk, tv, te):
        q, k, v, e, ws, wo = ctx.saved_tensors
        B, N, C = _attn_shapes(q, e)
        tq, tk, tv, te = _c(tq), _c(tk), _c(tv), _c(te)
        lib = _lib.load()
        gq, gk, gv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        ge = torch.empty_like(e)
        gws = torch.empty_like(e) if ws is not None else None
        gwo = torch.empty_like(q)
        with _dev(q):
            _lib.check(lib.dg_attn_core_bwd2(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(ws),
                                             _lib.ptr(wo), _lib.ptr(tq), _lib.ptr(tk), _lib.ptr(tv), _lib.ptr(te),
                                             _lib.ptr(gq), _lib.ptr(gk), _lib.ptr(gv), _lib.ptr(ge), _lib.ptr(gws),
                                             _lib.ptr(gwo), B, N, C, ctx.alpha, _lib.dt(q), _lib.stream_of(q)),
                       "dg_attn_core_bwd2")
        _account("attn_bwd2", q.element_size() * B * ((5 if ws is not None else 3) * N * N * C + 11 * N * C))
        return gq, gk, gv, ge, gws, gwo, None


def attn_core(q, k, v, e, alpha: float, need_s: bool = True):
    """(s, o) of the edge-modulated per-channel attention.

    s[b,i,j,c] = alpha q[b,i,c] k[b,j,c] (e^2+e)[b,i,j,c];  o = sum_j softmax_j(s) v_j.
    With ``need_s=False`` the [B,N,N,C] score tensor is not written (Discriminator's
    last block never reads it, reference models.py:202-207) and ``s`` is None.
    """
    s, o = _AttnCore.apply(q, k, v, e, float(alpha), bool(need_s))
    return (s if need_s else None), o


# --------------------------------------------------------------------------
# whole attention half of an Encoder_Block as one autograd node
# --------------------------------------------------------------------------
def _composite_attn_block(x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, b3, g4, b4, alpha, eps3,
                          eps4, need_edge):
    q, k, v = linear(x1, wq, bq), linear(x1, wk, bk), linear(x1, wv, bv)
    e = linear(y, we, be)
    s, o = attn_core(q, k, v, e, alpha, need_s=need_edge)
    x2 = linear_ln(o, won, bon, x1, g3, b3, eps3)
    if not need_edge:
        return x2
    return x2, linear_ln(s, woe, boe, y, g4, b4, eps4)


def attn_half_f32_supported(yf, N: int, C: int) -> bool:
    """dg_attn_half_f32_fwd (e projection + attention core + out_e + residual + ln4 as one float32 launch) serves C = 128 and
    row groups of at most 96 neighbours (above 48: two stages per row group, online softmax across them);
    options.attn_half_f32 = "off" keeps the three launches, "n48" keeps them above 48 neighbours (A/B measurements)."""
    mode = options.attn_half_f32
    return yf.is_cuda and yf.dtype == torch.float32 and C == 128 and N <= (48 if mode == "n48" else 96) and mode != "off"


def attn_half_f32_bwd1_supported(dy2f, B: int, N: int, C: int, graph: bool = False) -> bool:
    """dg_attn_half_f32_bwd1 (ln4 backward + out_e input gradient + attention-core backward as one float32 launch) serves
    C = 128 and row groups of at most 48 neighbours; its workgroups walk whole molecules, so it needs a batch that fills the
    chip (B >= 128; options.attn_half_f32_bwd = "force" lifts that for tests, "off" keeps the two launches, "nograph" keeps
    them only for passes a second order differentiates)."""
    mode = options.attn_half_f32_bwd
    return (dy2f.is_cuda and dy2f.dtype == torch.float32 and C == 128 and N <= 48 and mode != "off"
            and (B >= 128 or mode == "force") and (not graph or mode != "nograph"))


class _AttnBlock(Function):
    """x2 = LN3(x1 + out_n(o)), y2 = LN4(y + out_e(s)) with (s, o) = attention(q(x1), k(x1), v(x1), e(y))
    -- reference layers.py:111-135 + 186-190 -- as ONE autograd node: every projection is a row-GEMM
    launch with its bias / residual / LayerNorm epilogue, and in the backward every gradient
    accumulation (y feeds e-proj and the ln4 residual; x1 feeds q, k, v and the ln3 residual) is the
    residual operand of the next GEMM's epilogue instead of a separate elementwise add."""

    @staticmethod
    def forward(ctx, x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, b3, g4, b4, alpha, eps3, eps4,
                need_edge, ppre=None, pmean=None, prstd=None, pgamma=None, pbeta=None):
        # ppre .. pbeta: LNHandle of the LayerNorm that produced y (or None): its backward can then run in the
        # epilogue of this node's dy GEMM, the result leaving as the gradient of `ppre` instead of `y`
        B, N, C = x1.shape
        x1f, yf = _c(x1).reshape(-1, C), _c(y).reshape(-1, C)
        adt = x1f.dtype
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        if lin3_supported(x1f, (wq, wk, wv)):      # q, k, v share their input: one launch
            q, k, v = lin3(x1f, (wq, wk, wv), (bq, bk, bv))
        else:
            q = row_gemm(x1f, pw(wq, 0), C, C, bias=bq)
            k = row_gemm(x1f, pw(wk, 0), C, C, bias=bk)
            v = row_gemm(x1f, pw(wv, 0), C, C, bias=bv)
        lib = _lib.load()
        # no input needs a gradient (the Generator's forward inside the D step): the pre-LayerNorm sums are not written
        keep = any(ctx.needs_input_grad)
        o = torch.empty_like(q)
        fused_edge = need_edge and attn_half_f32_supported(yf, N, C)
        if fused_edge:
            # e projection, scores, softmax / node output, out_e, residual, ln4: one launch (dg_attn_half_f32_fwd); e, s and
            # the pre-LayerNorm sum are written for the backward only
            R = yf.shape[0]
            dev = yf.device
            e = torch.empty(R, C, dtype=adt, device=dev) if keep else None
            s = torch.empty(R, C, dtype=adt, device=dev) if keep else None
            y2 = torch.empty(R, C, dtype=adt, device=dev)
            pre4 = torch.empty(R, C, dtype=adt, device=dev) if keep else None
            mean4 = torch.empty(R, dtype=torch.float32, device=dev)
            rstd4 = torch.empty(R, dtype=torch.float32, device=dev)
            with _dev(q):
                _lib.check(lib.dg_attn_half_f32_fwd(_lib.ptr(yf), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), pw(we, 0).data_ptr(),
                                                    _lib.fptr(_c(be)), pw(woe, 0).data_ptr(), _lib.fptr(_c(boe)), _lib.fptr(_c(g4)),
                                                    _lib.fptr(_c(b4)), _lib.ptr(e), _lib.ptr(s), _lib.ptr(o), _lib.ptr(y2),
                                                    _lib.ptr(pre4), _lib.ptr(mean4), _lib.ptr(rstd4), B, N, C, alpha, eps4,
                                                    _lib.stream_of(q)), "dg_attn_half_f32_fwd")
            _account("attn_half_fwd", 4 * (R * C * (5 if keep else 2) + 4 * B * N * C), 4 * R * C * C,
                     floor=4 * (R * C * 2 + 4 * B * N * C))
        else:
            e = row_gemm(yf, pw(we, 0), C, C, bias=be)
            s = torch.empty_like(e) if need_edge else None
            with _dev(q):
                _lib.check(lib.dg_attn_core_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(s),
                                                _lib.ptr(o), B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)), "dg_attn_core_fwd")
            _account("attn_fwd", q.element_size() * B * ((2 if need_edge else 1) * N * N * C + 4 * N * C))
        r3 = row_gemm(o, pw(won, 0), C, C, bias=bon, residual=x1f, ln=(_c(g3), _c(b3), eps3), want_pre=keep)
        x2, mean3, rstd3, pre3 = r3 if keep else (*r3, None)
        outs = [x2.view(B, N, C)]
        # the penalty's forward: the parameters leave as alias outputs (see _weight_alias)
        ctx.alias = bool(keep and in_second_order_forward() and _alias_outputs_enabled())
        aliases = ()
        if ctx.alias:
            aliases = tuple(_weight_alias(t) for t in (wq, wk, wv, we, woe, won, g3, g4))
            wq, wk, wv, we, woe, won, g3, g4 = aliases
        saved = [x1, y, wq, wk, wv, we, woe, won, g3, g4, q, k, v, e, s, o, mean3, rstd3, pre3]
        if need_edge:
            if not fused_edge:
                r4 = row_gemm(s, pw(woe, 0), C, C, bias=boe, residual=yf, ln=(_c(g4), _c(b4), eps4), want_pre=keep)
                y2, mean4, rstd4, pre4 = r4 if keep else (*r4, None)
            outs.append(y2.view(B, N, N, C))
            saved += [mean4, rstd4, pre4]
        else:
            pre4 = None
        ctx.has_prev = ppre is not None
        if ctx.has_prev:
            saved += [ppre, pmean, prstd, pgamma]
        ctx.save_for_backward(*saved)
        ctx.cfg = (alpha, eps3, eps4, need_edge, (B, N, C))
        ctx.extra = (bq, bk, bv, be, boe, bon, b3, b4)
        ctx.set_materialize_grads(False)
        # pre3 / pre4 / q / k / v / e are outputs only so that the second order of the gradient penalty
        # can return their adjoints to THIS node, where they join the first-order gradients inside one
        # backward pass (see _AttnBlockBwd.backward); module code never sees them.
        return tuple(outs) + ((pre3, pre4) if need_edge else (pre3,)) + (q, k, v, e) + aliases

    @staticmethod
    def backward(ctx, dx2, *more):
        alpha, eps3, eps4, need_edge, (B, N, C) = ctx.cfg
        galias = ()
        if ctx.alias:      # second-order gradients of wq, wk, wv, we, woe, won, g3, g4 (or None each)
            more, galias = more[:-8], more[-8:]
        sv = ctx.saved_tensors
        x1, y, wq, wk, wv, we, woe, won, g3, g4, q, k, v, e, s, o, mean3, rstd3, pre3 = sv[:19]
        mean4, rstd4, pre4 = sv[19:22] if need_edge else (None, None, None)
        bq, bk, bv, be, boe, bon, b3, b4 = ctx.extra
        if need_edge:
            dy2, add3, add4, aq, ak, av, ae = more
            if dy2 is None:
                dy2 = torch.zeros_like(pre4)
        else:
            dy2 = add4 = None
            add3, aq, ak, av, ae = more
        if dx2 is None:
            dx2 = torch.zeros_like(pre3)
        wants_w = ctx.needs_input_grad[2] and not _inputs_only()
        want_aff = any(ctx.needs_input_grad[14:18]) and not _inputs_only()      # ln3 / ln4 affine parameters
        ppre = pmean = prstd = pgamma = None
        if ctx.has_prev:
            ppre, pmean, prstd, pgamma = sv[-4:]
        # y is the output of a LayerNorm whose handle came with it, and no graph is being recorded: that LayerNorm's
        # backward runs as the epilogue of the dy GEMM (its result is the gradient of `ppre`, y itself gets none)
        # (not in the last pass of a double backward -- recognisable by the adjoints of this node's extra outputs: there
        # the producing feed-forward node's `pre` ALSO receives the second-order adjoint, and autograd would join the
        # two with an edge-level add that costs more than the fused LayerNorm backward saves)
        second_pass = any(t is not None for t in (add3, add4, aq, ak, av, ae))
        fuse_prev = bool(ctx.has_prev and not torch.is_grad_enabled() and not second_pass and ctx.needs_input_grad[1]
                         and ctx.needs_input_grad[22] and row_gemm_ln_bwd_supported(q, C)
                         and tuple(ppre.shape) == (B * N * N, C))
        outs = _AttnBlockBwd.apply(x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, g4,
                                   q, k, v, e, s, o, mean3, rstd3, pre3, mean4, rstd4, pre4, dx2, dy2,
                                   add3, add4, aq, ak, av, ae,
                                   alpha, need_edge, ctx.needs_input_grad[0], ctx.needs_input_grad[1], wants_w,
                                   ppre if fuse_prev else None, pmean, prstd, pgamma, want_aff,
                                   not torch.is_grad_enabled())      # (no graph is being recorded: see _AttnBlockBwd)
        (dx1, dy, dwq, dbq, dwk, dbk, dwv, dbv, dwe, dbe, dwoe, dboe, dwon, dbon, dg3, db3, dg4, db4, dzp, dgp, dbp) = outs
        if any(g is not None for g in galias):
            dwq, dwk, dwv, dwe, dwoe, dwon, dg3, dg4 = _join_alias_grads((dwq, dwk, dwv, dwe, dwoe, dwon, dg3, dg4), galias)
        if not (ctx.needs_input_grad[25] and not _inputs_only()):
            dgp = dbp = None
        return (dx1, dy, dwq, dbq, dwk, dbk, dwv, dbv, dwe, dbe, dwoe, dboe, dwon, dbon, dg3, db3, dg4, db4,
                None, None, None, None, dzp, None, None, dgp, dbp)


def _attn_bwd_launch(q, k, v, e, ws, wo, alpha, add_e=None):
    B, N, C = q.shape[0], q.shape[1], q.shape[2]
    lib = _lib.load()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    de = torch.empty_like(e)
    with _dev(q):
        _lib.check(lib.dg_attn_core_bwd_add(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(ws),
                                            _lib.ptr(wo), _lib.ptr(add_e), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv),
                                            _lib.ptr(de), B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)),
                   "dg_attn_core_bwd")
    _account("attn_bwd", q.element_size() * B * ((2 + (ws is not None) + (add_e is not None)) * N * N * C + 7 * N * C))
    return dq, dk, dv, de


def _attn_bwd2_launch(q, k, v, e, ws, wo, tq, tk, tv, te, alpha):
    B, N, C = q.shape[0], q.shape[1], q.shape[2]
    lib = _lib.load()
    gq, gk, gv, gwo = (torch.empty_like(q) for _ in range(4))
    ge = torch.empty_like(e)
    gws = torch.empty_like(e) if ws is not None else None
    with _dev(q):
        _lib.check(lib.dg_attn_core_bwd2(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(e), _lib.ptr(ws), _lib.ptr(wo),
                                         _lib.ptr(tq), _lib.ptr(tk), _lib.ptr(tv), _lib.ptr(te), _lib.ptr(gq),
                                         _lib.ptr(gk), _lib.ptr(gv), _lib.ptr(ge), _lib.ptr(gws), _lib.ptr(gwo),
                                         B, N, C, alpha, _lib.dt(q), _lib.stream_of(q)), "dg_attn_core_bwd2")
    _account("attn_bwd2", q.element_size() * B * ((5 if ws is not None else 3) * N * N * C + 11 * N * C))
    return gq, gk, gv, ge, gws, gwo


class _AttnBlockBwd(Function):
    """Backward of ``_AttnBlock`` as a differentiable node; its own backward (second order of the
    gradient penalty) chains the same kernels: row GEMMs for every projection (forward packs where the
    first backward used the input-gradient packs and vice versa), dg_attn_core_bwd2 for the attention
    core, dg_ln_residual_bwd2 for ln3 / ln4.  The adjoints that reach the forward intermediates
    (pre-LayerNorm sums, q, k, v, e) are handed to the forward node as gradients of its extra outputs;
    they come back in as add3 / add4 / aq / ak / av / ae and are summed into the single first-order
    pass of that node."""

    @staticmethod
    def forward(ctx, *args):
        # one reduce launch for the block: six weight gradients + two LayerNorms' dgamma / dbeta
        wants_w = args[40] or (len(args) > 45 and args[45])      # weight gradients or LayerNorm affine gradients
        with _reduce_batch(args[0], on=bool(wants_w)) as inb:
            return _AttnBlockBwd._forward(ctx, inb, *args)

    @staticmethod
    def _forward(ctx, inb, x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, g4, q, k, v, e, s, o,
                 mean3, rstd3, pre3, mean4, rstd4, pre4, dx2, dy2, add3, add4, aq, ak, av, ae,
                 alpha, need_edge, want_x, want_y, wants_w, ppre=None, pmean=None, prstd=None, pgamma=None, want_aff=None,
                 no_graph=False):
        if want_aff is None:
            want_aff = wants_w
        B, N, C = x1.shape
        adt = q.dtype
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        cast = lambda t: t if t.dtype == adt else t.to(adt)
        x1f, yf = _c(x1).reshape(-1, C), _c(y).reshape(-1, C)
        dx2f = _c(cast(dx2)).reshape(-1, C)
        cadd = lambda t: None if t is None else _c(cast(t)).reshape(-1, C)
        dz3, dg3, db3 = _ln_bwd_rows(pre3, g3, mean3, rstd3, dx2f, cadd(add3), want_affine=want_aff,
                                     batch_slot=0 if inb else None)
        do = row_gemm(dz3, pw(won, 1), C, C).view(B, N, C)
        ds = dz4 = dg4 = db4 = dy2f = None
        qv, kv, vv, ev = q.view(B, N, C), k.view(B, N, C), v.view(B, N, C), e.view(B, N, N, C)
        fused1 = None
        if need_edge:
            dy2f = _c(cast(dy2)).reshape(-1, C)
            if (add4 is None and all(t is None for t in (aq, ak, av, ae))
                    and attn_half_f32_bwd1_supported(dy2f, B, N, C, graph=not no_graph)):
                # ln4 backward + ds = dz4 Woe + the attention core's backward: one launch; ds stays on chip unless a graph is
                # being recorded (the penalty's first backward: its second order reads ds)
                lib = _lib.load()
                dev = dy2f.device
                dz4, de = torch.empty_like(dy2f), torch.empty_like(dy2f)
                ds = None if no_graph else torch.empty_like(dy2f).view(B, N, N, C)
                dq, dk, dv = (torch.empty(B, N, C, dtype=adt, device=dev) for _ in range(3))
                dg4, db4 = (torch.empty(2, C, dtype=torch.float32, device=dev).unbind(0) if want_aff else (None, None))
                with _dev(dy2f):
                    ws = _scratch(dy2f, int(lib.dg_attn_half_f32_bwd1_workspace_bytes(B)), "ahb_batch" if inb else "ahb")
                    _lib.check(lib.dg_attn_half_f32_bwd1(_lib.ptr(dy2f), _lib.ptr(pre4), _lib.ptr(mean4), _lib.ptr(rstd4),
                                                         _lib.fptr(_c(g4)), pw(woe, 1).data_ptr(), _lib.ptr(ev), _lib.ptr(qv),
                                                         _lib.ptr(kv), _lib.ptr(vv), _lib.ptr(do), _lib.ptr(dz4), _lib.ptr(ds), _lib.ptr(de),
                                                         _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(dg4), _lib.ptr(db4),
                                                         ws.data_ptr(), ws.numel(), B, N, C, alpha, _lib.stream_of(dy2f)),
                               "dg_attn_half_f32_bwd1")
                _account("attn_half_bwd", 4 * (dy2f.shape[0] * C * (5 if ds is None else 6) + 7 * B * N * C), 2 * dy2f.shape[0] * C * C)
                de = de.view(B, N, N, C)
                fused1 = (dq, dk, dv, de)
            elif add4 is None and dy2f.shape[0] >= _lib.edge_rows() and ln_bwd_row_gemm_supported(dy2f, C, C):
                # ln4's backward runs in the producer waves of the out_e input-gradient GEMM (edge-level launches only:
                # at node level the three small launches it replaces are faster)
                dz4, ds, dg4, db4 = ln_bwd_row_gemm(pre4, g4, mean4, rstd4, dy2f, pw(woe, 1), want_affine=want_aff,
                                                    batch_slot=1 if inb else None)
                ds = ds.view(B, N, N, C)
            else:
                dz4, dg4, db4 = _ln_bwd_rows(pre4, g4, mean4, rstd4, dy2f, cadd(add4), want_affine=want_aff,
                                             batch_slot=1 if inb else None)
                ds = row_gemm(dz4, pw(woe, 1), C, C).view(B, N, N, C)
        # fp32: the adjoint of e joins de inside the kernel (one read stream instead of a 3-pass add).  The bf16
        # variant of that kernel is latency-bound at 2 waves / SIMD and the extra operand set costs more than the add
        # it saves (configs[2], A/B on one box: 217.2 vs 213.7 ms per step): bf16 adds afterwards.
        fold = ae is not None and adt == torch.float32
        aef = _c(cast(ae)).view(B, N, N, C) if fold else None      # joins de inside the kernel
        dq, dk, dv, de = fused1 if fused1 is not None else _attn_bwd_launch(qv, kv, vv, ev, ds, do, alpha, add_e=aef)
        pairs = [(got, extra.view(got.shape)) for got, extra in ((dq, aq), (dk, ak), (dv, av)) if extra is not None]
        if pairs:      # the node-level adjoints of q, k, v (second pass of the penalty): one multi-tensor launch
            torch._foreach_add_([g_ for g_, _ in pairs], [e_ if e_.dtype == g_.dtype else e_.to(g_.dtype) for g_, e_ in pairs])
        if ae is not None and not fold:
            de.add_(ae.view(de.shape))
        ctx.third = any(t is not None for t in (add3, add4, aq, ak, av, ae))
        dqf, dkf, dvf, def_ = dq.view(-1, C), dk.view(-1, C), dv.view(-1, C), de.view(-1, C)
        dy = dx1 = dzp = dgp = dbp = None
        if want_y and ppre is not None:      # + ln4 residual path, then the backward of the LayerNorm that made y
            dzp, dgp, dbp = row_gemm_ln_bwd(def_, pw(we, 1), C, dz4, ppre, pgamma, pmean, prstd)
        elif want_y:
            dy = row_gemm(def_, pw(we, 1), C, C, residual=dz4).view(y.shape)      # + ln4 residual path
        use3 = lin3_supported(dqf, (wq, wk, wv))      # dq Wq + dk Wk + dv Wv and the three weight gradients: one launch each
        if want_x and use3:
            dx1 = sum3(dqf, dkf, dvf, (wq, wk, wv), residual=dz3).view(x1.shape)   # + ln3 residual path
        elif want_x:
            t = row_gemm(dqf, pw(wq, 1), C, C, residual=dz3)                       # + ln3 residual path
            t = row_gemm(dkf, pw(wk, 1), C, C, residual=t)
            dx1 = row_gemm(dvf, pw(wv, 1), C, C, residual=t).view(x1.shape)
        gw = [None] * 12
        if wants_w:
            qkv_items = [((dqf, dkf, dvf), x1f, True)] if use3 else [(dqf, x1f, True), (dkf, x1f, True), (dvf, x1f, True)]
            items = qkv_items + [(def_, yf, True), (dz3, o, True)]
            if need_edge:
                items.append((dz4, s, True))
            # (out_n's weight gradient over the node rows rides in out_e's over the edge rows)
            res = _wgrad_many(items, open_batch=not inb, pair_from=len(items) - 2 if need_edge else None)
            if use3:      # rows 0..127 / 128..255 / 256..383 of the stacked gradient
                (w3, b3), res = res[0], res[1:]
                gw[0:6] = [w3[0:128], b3[0:128], w3[128:256], b3[128:256], w3[256:384], b3[256:384]]
            else:
                (gw[0], gw[1]), (gw[2], gw[3]), (gw[4], gw[5]) = res[:3]
                res = res[3:]
            (gw[6], gw[7]), (gw[10], gw[11]) = res[:2]
            if need_edge:
                gw[8], gw[9] = res[2]
        ctx.save_for_backward(x1, y, wq, wk, wv, we, woe, won, g3, g4, q, k, v, e, s, o, mean3, rstd3, pre3,
                              mean4, rstd4, pre4, dx2f, dy2f, dz3, dz4, do, ds, dq, dk, dv, de)
        ctx.cfg = (alpha, need_edge, (B, N, C), dx2.shape, None if dy2 is None else dy2.shape)
        ctx.set_materialize_grads(False)
        return (dx1, dy, *gw, dg3, db3, dg4, db4, dzp, dgp, dbp)

    @staticmethod
    @once_differentiable
    def backward(ctx, t1, ty, *rest):
        if any(r is not None for r in rest):
            raise RuntimeError("attn_block: second-order terms through parameter gradients are not implemented")
        if ctx.third:
            raise RuntimeError("attn_block: third-order differentiation is not implemented")
        alpha, need_edge, (B, N, C), dx2_shape, dy2_shape = ctx.cfg
        (x1, y, wq, wk, wv, we, woe, won, g3, g4, q, k, v, e, s, o, mean3, rstd3, pre3, mean4, rstd4, pre4,
         dx2f, dy2f, dz3, dz4, do, ds, dq, dk, dv, de) = ctx.saved_tensors
        adt = q.dtype
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        cast = lambda t: t if t.dtype == adt else t.to(adt)
        with_w = not _inputs_only()
        x1f, yf = _c(x1).reshape(-1, C), _c(y).reshape(-1, C)
        RN, RE = B * N, B * N * N
        zn = lambda: torch.zeros(RN, C, dtype=adt, device=q.device)
        t1f = _c(cast(t1)).reshape(-1, C) if t1 is not None else zn()
        tyf = _c(cast(ty)).reshape(-1, C) if ty is not None else torch.zeros(RE, C, dtype=adt, device=q.device)
        dqf, dkf, dvf, def_ = dq.view(-1, C), dk.view(-1, C), dv.view(-1, C), de.view(-1, C)
        # adjoints of dq, dk, dv, de (dx1 = dz3 + dq Wq + dk Wk + dv Wv ; dy = dz4 + de We)
        use3 = lin3_supported(t1f, (wq, wk, wv))
        if use3:
            tq, tk, tv = lin3(t1f, (wq, wk, wv), (None, None, None))
        else:
            tq = row_gemm(t1f, pw(wq, 0), C, C)
            tk = row_gemm(t1f, pw(wk, 0), C, C)
            tv = row_gemm(t1f, pw(wv, 0), C, C)
        te = row_gemm(tyf, pw(we, 0), C, C)
        qv, kv, vv, ev = q.view(B, N, C), k.view(B, N, C), v.view(B, N, C), e.view(B, N, N, C)
        gq, gk, gv, ge, gws, gwo = _attn_bwd2_launch(qv, kv, vv, ev, ds, do, tq.view(B, N, C), tk.view(B, N, C),
                                                     tv.view(B, N, C), te.view(B, N, N, C), alpha)
        # adjoints of dz3 / dz4 (do = dz3 Won, ds = dz4 Woe, plus the direct residual terms)
        adz3 = row_gemm(gwo.view(-1, C), pw(won, 0), C, C, residual=t1f)
        z3bar, dx2bar, g3bar = _ln_bwd2_rows(pre3, g3, mean3, rstd3, dx2f, adz3)
        z4bar = dy2bar = g4bar = None
        if need_edge:
            adz4 = row_gemm(gws.view(-1, C), pw(woe, 0), C, C, residual=tyf)
            z4bar, dy2bar, g4bar = _ln_bwd2_rows(pre4, g4, mean4, rstd4, dy2f, adz4)
        gW = [None] * 12
        if with_w:
            qkv_items = [((dqf, dkf, dvf), t1f, False)] if use3 else [(dqf, t1f, False), (dkf, t1f, False), (dvf, t1f, False)]
            items = qkv_items + [(def_, tyf, False), (dz3, gwo.view(-1, C), False)]
            if need_edge:
                items.append((dz4, gws.view(-1, C), False))
            res = _wgrad_many(items, pair_from=len(items) - 2 if need_edge else None)
            if use3:
                w3, res = res[0][0], res[1:]
                gW[0], gW[2], gW[4] = w3[0:128], w3[128:256], w3[256:384]
            else:
                gW[0], gW[2], gW[4] = (r[0] for r in res[:3])
                res = res[3:]
            gW[6], gW[10] = res[0][0], res[1][0]
            if need_edge:
                gW[8] = res[2][0]
        # The outputs depend on x1 / y only through the forward intermediates: their adjoints
        # (z3bar, z4bar at the pre-LayerNorm sums; gq, gk, gv, ge) go to the forward node.
        # inputs: x1, y, wq,bq, wk,bk, wv,bv, we,be, woe,boe, won,bon, g3, g4, q,k,v,e, s,o,
        #         mean3,rstd3,pre3, mean4,rstd4,pre4, dx2, dy2, 6 adds, 5 flags, 4 LNHandle fields, want_aff
        return (None, None, *gW, g3bar, g4bar, gq.view_as(q), gk.view_as(k), gv.view_as(v), ge.view_as(e), None, None,
                None, None, z3bar, None, None, z4bar, dx2bar.view(dx2_shape),
                None if dy2bar is None else dy2bar.view(dy2_shape), *([None] * 17))


_half_pack_cache = {}


def _attn_half_packed(we, woe, dtype):
    """Fragment-order copies of (e.weight, out_e.weight) and their transposes for the fused attention-half kernels
    (dg_attn_half_pack), cached like ``packed_weight``."""
    key = (id(we), id(woe), dtype)
    hit = _half_pack_cache.get(key)
    if (hit is not None and hit[0]() is we and hit[1]() is woe and hit[2] == (we._version, woe._version)
            and hit[4] == (we.data_ptr(), woe.data_ptr()) and hit[5] == _weights_epoch[0]):
        return hit[3]
    if len(_half_pack_cache) > 1024:
        for k in [k for k, v in _half_pack_cache.items() if v[0]() is None or v[1]() is None]:
            del _half_pack_cache[k]
    lib = _lib.load()
    code = _lib.DTYPES[dtype]
    packed = torch.empty(int(lib.dg_attn_half_packed_bytes(code)), dtype=torch.uint8, device=we.device)
    with _dev(we):
        _lib.check(lib.dg_attn_half_pack(_lib.fptr(_c(we.detach())), _lib.fptr(_c(woe.detach())), packed.data_ptr(), code,
                                         _lib.stream_of(we)), "dg_attn_half_pack")
    _half_pack_cache[key] = (weakref.ref(we), weakref.ref(woe), (we._version, woe._version), packed,
                             (we.data_ptr(), woe.data_ptr()), _weights_epoch[0])
    return packed


def _fused_attn_half_enabled() -> bool:
    """options.attn_half = "unfused" keeps the bf16 attention half on the separate launches (A/B measurements)."""
    return options.attn_half != "unfused"


def attn_half_supported(dtype, N: int, C: int) -> bool:
    """Shapes the module path routes to the fused attention-half kernels.  The kernels accept N <= 96, but above 48 the
    backward keeps 6 row blocks of accumulators per lane and spills (N = 90, B = 64: 633 vs 645 molecules/s for the
    separate launches), so BASELINE configs[4] stays on those; options.attn_half = "force" routes every N <= 96 (tests)."""
    limit = 96 if options.attn_half == "force" else 48
    return dtype == torch.bfloat16 and C == 128 and 1 <= N <= limit


class _AttnBlockFused(Function):
    """The same block as ``_AttnBlock`` with the whole EDGE side -- e-projection, Hadamard score, softmax over j, AV,
    out_e, residual, ln4 (reference layers.py:116-135,186-190) -- in ONE kernel per direction (csrc/attn_half.hip):
    ``e`` and ``s`` never exist in HBM, the backward recomputes them from the saved layer input ``y`` and accumulates
    the weight gradients of e / out_e inside the kernel.  The node side (q, k, v, out_n + ln3; R = B N rows) stays on
    the row GEMMs.  First order only: graphs that will be differentiated twice are built from ``_AttnBlock``."""

    @staticmethod
    def forward(ctx, x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, b3, g4, b4, alpha, eps3, eps4,
                need_edge):
        B, N, C = x1.shape
        x1f = _c(x1).reshape(-1, C)
        yc = _c(y)
        adt = x1f.dtype
        code = _lib.dt(x1f)
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        q = row_gemm(x1f, pw(wq, 0), C, C, bias=bq)
        k = row_gemm(x1f, pw(wk, 0), C, C, bias=bk)
        v = row_gemm(x1f, pw(wv, 0), C, C, bias=bv)
        lib = _lib.load()
        dev = x1f.device
        o = torch.empty_like(q)
        y2 = pre4 = mean4 = rstd4 = None
        if need_edge:
            y2 = torch.empty_like(yc)
            pre4 = torch.empty_like(yc)
            mean4 = torch.empty(B * N * N, dtype=torch.float32, device=dev)
            rstd4 = torch.empty(B * N * N, dtype=torch.float32, device=dev)
        packed = _attn_half_packed(we, woe, adt)
        with _dev(q):
            _lib.check(lib.dg_attn_half_fwd(_lib.ptr(yc), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), packed.data_ptr(),
                                            _lib.fptr(_c(be)), _lib.fptr(_c(boe)), _lib.fptr(_c(g4)), _lib.fptr(_c(b4)),
                                            _lib.ptr(o), _lib.ptr(y2), _lib.ptr(pre4), _lib.ptr(mean4), _lib.ptr(rstd4),
                                            B, N, C, alpha, eps4, code, _lib.stream_of(q)), "dg_attn_half_fwd")
        es = q.element_size()
        _account("attn_half_fwd", es * B * ((3 if need_edge else 1) * N * N * C + 4 * N * C),
                 2 * B * N * N * C * C * (2 if need_edge else 1), floor=es * B * ((2 if need_edge else 1) * N * N * C + 4 * N * C))
        x2, mean3, rstd3, pre3 = row_gemm(o, pw(won, 0), C, C, bias=bon, residual=x1f, ln=(_c(g3), _c(b3), eps3),
                                          want_pre=True)
        ctx.save_for_backward(x1, yc, wq, wk, wv, we, woe, won, g3, g4, be, q, k, v, o, mean3, rstd3, pre3, mean4, rstd4,
                              pre4)
        ctx.cfg = (alpha, need_edge, (B, N, C))
        ctx.extra = (bq, bk, bv, boe, bon, b3, b4, eps3, eps4)
        ctx.set_materialize_grads(False)
        if need_edge:
            return x2.view(B, N, C), y2
        return x2.view(B, N, C)

    @staticmethod
    def backward(ctx, dx2, dy2=None):
        wants_w = bool((ctx.needs_input_grad[2] or any(ctx.needs_input_grad[14:18])) and not _inputs_only()
                       and not torch.is_grad_enabled())
        with _reduce_batch(ctx.saved_tensors[0], on=wants_w) as inb:      # one reduce launch for the block
            return _AttnBlockFused._backward(ctx, inb, dx2, dy2)

    @staticmethod
    def _backward(ctx, inb, dx2, dy2=None):
        alpha, need_edge, (B, N, C) = ctx.cfg
        (x1, y, wq, wk, wv, we, woe, won, g3, g4, be, q, k, v, o, mean3, rstd3, pre3, mean4, rstd4,
         pre4) = ctx.saved_tensors
        if torch.is_grad_enabled():      # create_graph=True outside second_order_forward(): composite graph
            bq, bk, bv, boe, bon, b3, b4, eps3, eps4 = ctx.extra
            ins = (x1, y, wq, bq, wk, bk, wv, bv, we, be, woe, boe, won, bon, g3, b3, g4, b4)
            gouts = (dx2, dy2) if need_edge else dx2
            return _double_backward_fallback(
                lambda *t: _composite_attn_block(*t, alpha, eps3, eps4, need_edge), ins, gouts) + (None,) * 4
        adt = q.dtype
        code = _lib.dt(q)
        dev = q.device
        pw = lambda w_, m_: packed_weight(w_, m_, adt)
        cast = lambda t: t if t.dtype == adt else t.to(adt)
        wants_w = ctx.needs_input_grad[2] and not _inputs_only()
        want_aff = any(ctx.needs_input_grad[14:18]) and not _inputs_only()
        x1f = _c(x1).reshape(-1, C)
        if dx2 is None:
            dx2 = torch.zeros_like(pre3)
        dz3, dg3, db3 = _ln_bwd_rows(pre3, g3, mean3, rstd3, _c(cast(dx2)).reshape(-1, C), want_affine=want_aff,
                                     batch_slot=0 if inb else None)
        do = row_gemm(dz3, pw(won, 1), C, C)
        dz4 = dg4 = db4 = None
        if need_edge:
            if dy2 is None:
                dy2 = torch.zeros_like(pre4)
            dz4, dg4, db4 = _ln_bwd_rows(pre4.view(-1, C), g4, mean4, rstd4, _c(cast(dy2)).reshape(-1, C), want_affine=want_aff,
                                         batch_slot=1 if inb else None)
        lib = _lib.load()
        dy = torch.empty_like(y)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        dwe = dbe = dwoe = dboe = None
        if wants_w:
            dwe = torch.empty_like(we)
            dbe = torch.empty(C, dtype=torch.float32, device=dev)
            if need_edge:
                dwoe = torch.empty_like(woe)
                dboe = torch.empty(C, dtype=torch.float32, device=dev)
        need = int(lib.dg_attn_half_bwd_workspace_bytes(B, N))
        with _dev(q):
            ws = _scratch(q, need, "half")
            _lib.check(lib.dg_attn_half_bwd(_lib.ptr(y), _lib.ptr(dz4), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(do),
                                            _attn_half_packed(we, woe, adt).data_ptr(), _lib.fptr(_c(be)), _lib.ptr(dy),
                                            _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(dwe), _lib.ptr(dbe),
                                            _lib.ptr(dwoe), _lib.ptr(dboe), ws.data_ptr(), ws.numel(), B, N, C, alpha,
                                            code, _lib.stream_of(q)), "dg_attn_half_bwd")
        es = q.element_size()
        _account("attn_half_bwd", es * B * ((3 if need_edge else 2) * N * N * C + 8 * N * C),
                 2 * B * N * N * C * C * ((3 if need_edge else 2) + (2 if wants_w and need_edge else (1 if wants_w else 0))))
        dx1 = None
        if ctx.needs_input_grad[0]:
            t = row_gemm(dq, pw(wq, 1), C, C, residual=dz3)                        # + ln3 residual path
            t = row_gemm(dk, pw(wk, 1), C, C, residual=t)
            dx1 = row_gemm(dv, pw(wv, 1), C, C, residual=t).view(x1.shape)
        gw = [None] * 12
        if wants_w:
            (gw[0], gw[1]), (gw[2], gw[3]), (gw[4], gw[5]), (gw[10], gw[11]) = _wgrad_many(
                [(dq, x1f, True), (dk, x1f, True), (dv, x1f, True), (dz3, o, True)], open_batch=not inb)
            gw[6], gw[7] = dwe, dbe
            gw[8], gw[9] = dwoe, dboe
        return (dx1, (dy if ctx.needs_input_grad[1] else None), *gw, dg3, db3, dg4, db4, None, None, None, None)


def attn_block(x1, y, attn, ln3, ln4, need_edge=True, y_ln=None):
    """Attention half of an encoder block for ``attn`` (an MHA module): returns
    (LN3(x1 + out_n(o)), LN4(y + out_e(s)) or None).  ``y_ln``: the LNHandle of the LayerNorm whose output y is
    (``ffn_ln(..., want_handle=True)``), or None."""
    C = x1.shape[-1]
    alpha = 1.0 / (attn.d_k ** 0.5)
    args = (x1, y, attn.q.weight, attn.q.bias, attn.k.weight, attn.k.bias, attn.v.weight, attn.v.bias,
            attn.e.weight, attn.e.bias, attn.out_e.weight, attn.out_e.bias, attn.out_n.weight, attn.out_n.bias,
            ln3.weight, ln3.bias, ln4.weight, ln4.bias)
    fused = (x1.is_cuda and x1.dtype in _lib.DTYPES and y.dtype == x1.dtype and C == 128 and x1.dim() == 3
             and all(t is not None for t in args))
    if not fused:
        out = _composite_attn_block(*args, alpha, ln3.eps, ln4.eps, need_edge)
    elif (not in_second_order_forward() and attn_half_supported(x1.dtype, x1.shape[1], C) and _fused_attn_half_enabled()
          and tuple(y.shape) == (x1.shape[0], x1.shape[1], x1.shape[1], C)):
        out = _AttnBlockFused.apply(*args, alpha, ln3.eps, ln4.eps, need_edge)
        return (out[0], out[1]) if need_edge else (out, None)
    else:
        prev = (None,) * 5 if y_ln is None else (y_ln.pre, y_ln.mean, y_ln.rstd, y_ln.gamma, y_ln.beta)
        out = _AttnBlock.apply(*args, alpha, ln3.eps, ln4.eps, need_edge, *prev)
        return (out[0], out[1]) if need_edge else (out[0], None)
    return out if need_edge else (out, None)


__all__ = [_n for _n in dir() if not _n.startswith("__")]
