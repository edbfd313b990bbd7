"""One WGAN-GP iteration, data-parallel over molecules.

``GANStep.step`` restates the reference's inner loop (``train.py:351-384``):

    reset_grad -> discriminator_loss -> backward -> d_optimizer.step
    reset_grad -> generator_loss     -> backward -> g_optimizer.step

without the per-step ``.item()`` host syncs (``train.py:364-366,380-382``).

Multi-GPU: one process per GPU, each rank owns an equal
shard of the molecule batch; every loss term is a batch mean, so the global
gradient is the average of the rank gradients.  The reference's
``nn.DataParallel`` (``train.py:220-223``: broadcast + scatter + gather +
reduce_add every forward) is replaced by ONE flat-bucket all-reduce per
backward -- D gradients after the D backward, G gradients after the G backward
-- over RCCL/xGMI (backend "nccl" on ROCm).  There is no other collective on
the data path.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import os

import torch
import torch.distributed as dist

from . import _lib
from .model.loss import discriminator_loss, generator_loss
from .functional import as_one_hot, attach_one_hot_labels, bump_weights_epoch, one_hot_labels
from .options import options
from .optim import FlatAdamW

__all__ = ["GANStep", "GraphedGANStep", "GradBucket", "broadcast_parameters"]


def broadcast_parameters(module: torch.nn.Module, group=None, src: int = 0) -> None:
    """Make every rank start from rank ``src``'s weights (done once)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
        if not tensors:
            return
        flat = torch._utils._flatten_dense_tensors(tensors)
        dist.broadcast(flat, src=src, group=group)
        for t, f in zip(tensors, torch._utils._unflatten_dense_tensors(flat, tensors)):
            t.copy_(f)


class GradBucket:
    """Flat gradient bucket of one network: all live ``.grad`` tensors are
    averaged across ranks with a single all-reduce.

    Parameters whose ``.grad`` is None stay None (the Discriminator's dead
    last-block edge branch, reference models.py:202-207): the set of live
    parameters is structural, hence identical on every rank.
    """

    def __init__(self, module: torch.nn.Module, group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.group = group
        self._flat: Optional[torch.Tensor] = None
        self._live: Optional[Tuple[int, ...]] = None

    def world_size(self) -> int:
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def all_reduce_mean(self) -> None:
        ws = self.world_size()
        if ws == 1:
            return
        live = tuple(i for i, p in enumerate(self.params) if p.grad is not None)
        if not live:
            return
        grads = [self.params[i].grad for i in live]
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._live != live or self._flat.numel() != n or self._flat.device != grads[0].device:
            self._flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
            self._live = live
        views, off = [], 0
        for g in grads:
            views.append(self._flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        torch._foreach_copy_(views, grads)
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            dist.all_reduce(self._flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
            self._flat.div_(ws)
        torch._foreach_copy_(grads, views)


class GANStep:
    """Owns the two AdamW optimizers (``train.py:213-214``) and runs iterations."""

    def __init__(self, G: torch.nn.Module, D: torch.nn.Module, *, g_lr: float = 1e-5, d_lr: float = 1e-5,
                 betas: Sequence[float] = (0.9, 0.999), lambda_gp: float = 10.0, group=None,
                 skip_d_wgrad_in_g_step: bool = True, d_loss_fn=discriminator_loss, g_loss_fn=generator_loss,
                 optimizer: str = "auto", share_generator_forward: bool = True, memory: str = "auto",
                 split_d_backward: bool = True):
        self.G, self.D = G, D
        # the critic terms and the penalty are differentiated by two backward passes joined by ONE multi-tensor add (False: the
        # reference's single backward over their sum, for users of per-parameter hooks on D)
        self.split_d_backward = split_d_backward
        self.lambda_gp = lambda_gp
        self.group = group
        on_gpu = next(G.parameters()).is_cuda
        if optimizer == "flat" or (optimizer == "auto" and on_gpu):
            # one AdamW kernel per network over a flat buffer that doubles as the all-reduce bucket
            self.g_optimizer = FlatAdamW(G.parameters(), g_lr, tuple(betas))
            self.d_optimizer = FlatAdamW(D.parameters(), d_lr, tuple(betas))
        else:
            self.g_optimizer = torch.optim.AdamW(G.parameters(), g_lr, tuple(betas))
            self.d_optimizer = torch.optim.AdamW(D.parameters(), d_lr, tuple(betas))
        self.g_bucket, self.d_bucket = GradBucket(G, group), GradBucket(D, group)
        # The reference also computes D's weight gradients in the G step and throws
        # them away at the next reset_grad (train.py:352); skipping them changes nothing
        # observable and saves ~2.2e9 FLOP per molecule.
        self.skip_d_wgrad_in_g_step = skip_d_wgrad_in_g_step
        # The reference runs G(mol) twice per iteration -- loss.py:60 for the D step (detached) and
        # loss.py:77 for the G step -- with the SAME generator weights and inputs (G is only updated
        # at the end of the iteration, train.py:384).  Without active dropout the two forwards are
        # identical, so one forward with its graph serves both.
        self.share_generator_forward = share_generator_forward
        self._d_loss_fn, self._g_loss_fn = d_loss_fn, g_loss_fn
        # memory = "fast": one backward over the whole D loss, the generator forward shared by both steps (three
        # Discriminator graphs and the Generator graph alive at once: ~1.2 GB per molecule at N=45, L=4 in fp32);
        # "low": the three D terms are differentiated one after the other (-mean D(real), mean D(fake), lambda gp:
        # gradients accumulate, each graph is freed before the next is built) and G runs again for the G step --
        # same numbers up to the fp32 order of three additions per gradient element, a third of the peak memory.
        # "auto" picks "low" when the fast path's estimate does not fit the device (BASELINE configs[3]: B = 2048
        # per GPU in fp32 needs > 288 GB on the fast path).
        if memory not in ("auto", "fast", "low"):
            raise ValueError("memory must be 'auto', 'fast' or 'low'")
        if memory == "low" and (d_loss_fn is not discriminator_loss or g_loss_fn is not generator_loss):
            raise ValueError("memory='low' differentiates the terms of the default discriminator_loss one by one: it "
                             "cannot run a custom d_loss_fn / g_loss_fn (use memory='fast' or 'auto')")
        self.memory = memory
        # bench.py --gpus N: HIP events around the two gradient all-reduces of every step ((start, stop) pairs on the
        # launch stream; None = off)
        self.collective_events = None
        self._reduce_hook = None      # set by GraphedGANStep while it captures a data-parallel step in segments

    def time_collectives(self, on: bool = True) -> None:
        """Start (or stop) recording a pair of HIP events around every gradient all-reduce."""
        self.collective_events = [] if on else None

    def collective_ms(self):
        """(all-reduces timed, total ms between their events) since ``time_collectives()`` or the previous call; synchronises
        the events.  The events sit on the current HIP stream: with the `nccl` backend the span is the device-side collective,
        with `gloo` it is the host-side copy + reduction the stream waits for."""
        ev = self.collective_events or []
        if ev:
            ev[-1][1].synchronize()
        n, ms = len(ev), float(sum(a.elapsed_time(b) for a, b in ev))
        if self.collective_events is not None:      # read once: the events are released, a long run does not accumulate them
            self.collective_events = []
        return n, ms

    def _low_memory(self, gen_edge) -> bool:
        if self.memory != "auto":
            return self.memory == "low"
        if not gen_edge.is_cuda or self._d_loss_fn is not discriminator_loss or self._g_loss_fn is not generator_loss:
            return False
        from .functional import activation_dtype
        B, N = gen_edge.shape[0], gen_edge.shape[1]
        dim = int(getattr(self.G, "dim", 128))
        # block passes whose graphs are alive at the peak: G once, D four times (real + fake, gradient penalty
        # and its double backward)
        depth = int(getattr(self.G, "depth", 1)) + 4 * int(getattr(self.D, "depth", 1))
        es = 2 if activation_dtype() == torch.bfloat16 else 4
        # measured on MI355X at N=45, L=4 (bench.py peak_memory_GB): 0.18 GB per molecule in fp32, 0.077 GB with bf16
        # activations = 8.7 / 7.4 edge tensors per block pass; 9 leaves a margin
        estimate = 9.0 * B * N * N * dim * es * depth
        return estimate > 0.85 * torch.cuda.get_device_properties(gen_edge.device).total_memory

    def _low_memory_shares_generator(self, gen_edge) -> bool:
        """Low-memory step: may the ONE generator forward (with its graph) still serve both steps?  The peak of the low-memory
        step is two Discriminator block-pass sets (the penalty's graph and its double backward: 154 GB at B = 2048, N = 45,
        L = 4 in fp32); keeping G's graph alive through the D step adds one of G's (76 GB there)."""
        if not (self.share_generator_forward and self._d_loss_fn is discriminator_loss and self._g_loss_fn is generator_loss):
            return False
        if self.G.training and float(getattr(self.G, "dropout", 0.0) or 0.0) > 0.0:
            return False
        mode = options.low_memory_share
        if mode != "auto":
            return mode == "on"
        from .functional import activation_dtype
        B, N = gen_edge.shape[0], gen_edge.shape[1]
        dim = int(getattr(self.G, "dim", 128))
        es = 2 if activation_dtype() == torch.bfloat16 else 4
        unit = 9.0 * B * N * N * dim * es
        estimate = unit * (2 * int(getattr(self.D, "depth", 1)) + int(getattr(self.G, "depth", 1)))
        # (0.75: the estimate is rough -- B = 2048, N = 45, L = 4 in fp32 measures 230 GB against an estimate of 229 -- and a
        # fragmented allocator or a slightly larger N must fall back to the second generator forward, not run out of memory)
        return estimate <= 0.75 * torch.cuda.get_device_properties(gen_edge.device).total_memory

    def _d_step_low_memory(self, disc_edge, disc_node, gen_edge, gen_node, B, dev, eps, samples=None):
        """discriminator_loss (reference loss.py:52-72) + backward, one term at a time."""
        from .model.loss import gradient_penalty
        if samples is not None:
            node_sample, edge_sample = samples
        else:
            with torch.no_grad():
                _, _, node_sample, edge_sample = self.G(gen_edge, gen_node)
        loss_real = -torch.mean(self.D(disc_edge, disc_node))
        loss_real.backward()
        loss_fake = torch.mean(self.D(edge_sample, node_sample))
        loss_fake.backward()
        gp = self.lambda_gp * gradient_penalty(self.D, disc_node, disc_edge, node_sample, edge_sample, B, dev, eps=eps)
        gp.backward()
        return (loss_fake.detach() + loss_real.detach() + gp.detach())

    def _reduce_flat(self, flat: torch.Tensor, ws: int) -> None:
        """Average the flat gradient bucket across ranks: ONE all-reduce (timed by a pair of HIP events when asked)."""
        timed = self.collective_events is not None and flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(ws)
        if timed:
            e1.record()
            if len(self.collective_events) < 4096:      # bounded: timing left on for a long run keeps the first 4096
                self.collective_events.append((e0, e1))

    def _update(self, opt, bucket: GradBucket) -> None:
        """Average gradients across ranks (one all-reduce) and apply AdamW."""
        if isinstance(opt, FlatAdamW):
            flat = opt.pack_grads()
            if flat is None:
                return
            ws = bucket.world_size()
            if self._reduce_hook is not None:      # GraphedGANStep's segmented capture: the collective stays outside the graphs
                self._reduce_hook(flat, ws)
            elif ws > 1:
                self._reduce_flat(flat, ws)
            opt.step(packed=True)
        else:
            bucket.all_reduce_mean()
            opt.step()

    def reset_grad(self) -> None:
        self.g_optimizer.zero_grad(set_to_none=True)
        self.d_optimizer.zero_grad(set_to_none=True)

    def step(self, disc_edge, disc_node, gen_edge, gen_node, eps=None):
        """One iteration on this rank's shard.  Returns (d_loss, g_loss) as
        0-dim device tensors (local-shard values; no host sync)."""
        # edge-level = at least half of this shard's B N^2 edge rows (profiler keys, traversal direction, the LayerNorm-backward
        # prologue's level test): a large batch's node-level launches (B = 2048: 92 160 / 184 320 rows) must not be filed with
        # the edge-level ones.  Scoped to the step: the previous threshold comes back when it returns (a later, smaller
        # forward / backward in the same process is classified by ITS caller again).
        prev_rows = _lib.edge_rows()
        _lib.set_edge_rows(max(_lib.EDGE_ROWS, gen_node.shape[0] * gen_node.shape[1] ** 2 // 2))
        try:
            return self._step(disc_edge, disc_node, gen_edge, gen_node, eps)
        finally:
            _lib.set_edge_rows(prev_rows)

    def _step(self, disc_edge, disc_node, gen_edge, gen_node, eps=None):
        B, dev = gen_node.shape[0], gen_node.device
        # dataset graphs are one-hot (reference utils.py:15-23): checked once per tensor object, then the edge
        # embedding of these two batches is a table gather instead of an MLP over B N^2 rows
        gen_edge, disc_edge = as_one_hot(gen_edge), as_one_hot(disc_edge)
        self.reset_grad()
        kw = {} if eps is None else {"eps": eps}
        shared = None
        low = self._low_memory(gen_edge)
        if low:
            samples = None
            if self._low_memory_shares_generator(gen_edge):      # one generator forward for both steps, as in the fast step
                shared = self.G(gen_edge, gen_node)
                samples = (shared[2].detach(), shared[3].detach())
            d_loss = self._d_step_low_memory(disc_edge, disc_node, gen_edge, gen_node, B, dev, eps, samples)
        elif (self.share_generator_forward and self._d_loss_fn is discriminator_loss
                and self._g_loss_fn is generator_loss
                and not (self.G.training and float(getattr(self.G, "dropout", 0.0) or 0.0) > 0.0)):
            shared = self.G(gen_edge, gen_node)
            kw["generator_outputs"] = shared
        if not low and self._d_loss_fn is discriminator_loss and self.split_d_backward:
            # The critic terms and the penalty share nothing but D's parameters.  One backward over their sum makes the
            # autograd engine add the two contributions of every parameter with a kernel of its own (~140 tiny adds per
            # step); two backward passes and ONE multi-tensor add give the same sums in the same order.
            _, _, d_loss, main, pen = discriminator_loss(self.G, self.D, disc_edge, disc_node, gen_edge, gen_node, B, dev,
                                                         self.lambda_gp, return_terms=True, **kw)
            main.backward()
            params = [p for p in self.D.parameters() if p.requires_grad]
            # (torch.autograd.grad: tensor / post-accumulate-grad hooks on D's parameters fire for `main` only, not for
            # the penalty term; GANStep(split_d_backward=False) runs the reference's single backward for hook users)
            g2 = torch.autograd.grad(pen, params, allow_unused=True)
            have, add = [], []
            for p, g in zip(params, g2):
                if g is None:
                    continue
                if p.grad is None:
                    p.grad = g
                else:
                    have.append(p.grad)
                    add.append(g)
            if have:
                torch._foreach_add_(have, add)
        elif not low:
            _, _, d_loss = self._d_loss_fn(self.G, self.D, disc_edge, disc_node, gen_edge, gen_node, B, dev,
                                           self.lambda_gp, **kw)
            d_loss.backward()
        self._update(self.d_optimizer, self.d_bucket)
        self.reset_grad()
        d_params = [p for p in self.D.parameters() if p.requires_grad] if self.skip_d_wgrad_in_g_step else []
        for p in d_params:
            p.requires_grad_(False)
        try:
            gkw = {} if shared is None else {"generator_outputs": shared}
            g_loss = self._g_loss_fn(self.G, self.D, gen_edge, gen_node, B, **gkw)[0]
            g_loss.backward()
        finally:
            for p in d_params:
                p.requires_grad_(True)
        self._update(self.g_optimizer, self.g_bucket)
        return d_loss.detach(), g_loss.detach()


class GraphedGANStep:
    """The whole iteration (both forwards, the gradient penalty with its double backward, both
    backwards and both AdamW updates: ~2.5 k kernel launches) captured once into a hipGraph and
    replayed.  For small molecules / batches the eager step is launch-bound (BASELINE configs[0]
    shape: 13.7 ms eager -> 3.3 ms replayed on MI355X); at configs[1] the GPU is already saturated.

    One process: ONE graph.  Under data parallelism (one process per GPU, world size > 1) a collective cannot sit in the
    graph, so the iteration is captured as THREE graphs that share one memory pool, cut at the two gradient all-reduces
    (the minimum the alternating updates allow) --
        [G forward, D loss, its backward, flat D gradient bucket] all-reduce [AdamW(D), G loss, its backward, flat G bucket]
        all-reduce [AdamW(G)]
    -- and a replay runs graph, eager all-reduce, graph, eager all-reduce, graph: two collectives and three launches from
    the host per step instead of ~750 (a rank whose Python is slow no longer delays the collective of all).  The autograd
    graph of the shared generator forward is built during the first capture and consumed during the second: legal because
    the captures share the pool and replay in capture order.  New batches are copied into the static input
    buffers; ``eps`` is drawn on the device inside the graph like the reference does.

    One-hot edge batches (dataset graphs) are embedded through their int32 LABELS (table gather): the
    captured kernels read static label buffers owned by this object, and ``step`` refreshes them together
    with the dense buffers -- a replay never sees the labels of an older batch."""

    def __init__(self, stepper: GANStep, disc_edge, disc_node, gen_edge, gen_node, warmup: int = 3, eps=None, segmented=None):
        # eps: (eps_edge [B,1,1,1], eps_node [B,1,1]) of the gradient penalty's interpolation as STATIC inputs (``step(eps=...)``
        # refreshes them) instead of the uniform draw inside the graph -- reproducible runs, tests
        # segmented: None = three graphs exactly when the process group has more than one rank; True forces them (a one-rank
        # group then still issues its two all-reduces between the graphs: tests of the RCCL call sequence on one GPU)
        self.static_eps = None if eps is None else tuple(t.clone() for t in eps)
        self.world = (dist.get_world_size(stepper.group) if dist.is_available() and dist.is_initialized() else 1)
        if self.world > 1 and not isinstance(stepper.d_optimizer, FlatAdamW):
            raise RuntimeError("GraphedGANStep under data parallelism needs the flat optimizer (its gradient buffer is the "
                               "all-reduce bucket the graphs are cut around)")
        self.stepper = stepper
        self.static = [t.clone() for t in (disc_edge, disc_node, gen_edge, gen_node)]
        # static label buffers of the two edge batches (index 0: D's real batch, 2: the generator input): validated
        # HERE, outside the graph (as_one_hot syncs once); None when the capture batch is not one-hot -- the graph then
        # records the dense embedding kernels and needs no labels
        self._labels = {}
        for idx, src in ((0, disc_edge), (2, gen_edge)):
            lab = one_hot_labels(as_one_hot(src))
            if lab is not None:
                self._labels[idx] = lab.clone()
                attach_one_hot_labels(self.static[idx], self._labels[idx])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the default stream: allocator pools, packed
            for _ in range(warmup):            # weights, flat optimizer state, hipFuncSetAttribute calls
                stepper.step(*self.static, eps=self.static_eps)
        torch.cuda.current_stream().wait_stream(side)
        # Packed GEMM weights are cached per (tensor, version, epoch).  Without this bump the capture
        # below would hit the warm-up's packs for every weight that is not updated before its first
        # use inside the step: no pack kernel would be recorded and every replay would run D's GEMMs
        # on weights frozen at the end of warm-up.  With it, each first use records its pack kernel,
        # so a replay re-packs from the live parameters exactly like an eager step.
        bump_weights_epoch()
        self.graph = torch.cuda.CUDAGraph()
        self.segments = None
        if self.world == 1 and not segmented:
            with torch.cuda.graph(self.graph):
                self.losses = stepper.step(*self.static, eps=self.static_eps)
        else:
            self._capture_segments()
        bump_weights_epoch()    # cache entries made during capture point into the graph's private pool

    def _capture_segments(self):
        """Capture one step as consecutive graphs, a new one starting wherever GANStep._update would all-reduce.  Nothing
        runs during a capture, so no collective is issued here: every rank captures on its own."""
        graphs, flats = [self.graph], []
        pool = torch.cuda.graph_pool_handle()
        stepper = self.stepper

        def cut(flat, ws):      # called by GANStep._update between pack_grads() and the optimizer step
            graphs[-1].capture_end()
            flats.append(flat)
            graphs.append(torch.cuda.CUDAGraph())
            graphs[-1].capture_begin(pool=pool, capture_error_mode="thread_local")

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        stepper._reduce_hook = cut
        try:
            with torch.cuda.stream(side):
                # thread_local: the process group's watchdog thread polls events while this thread captures
                graphs[0].capture_begin(pool=pool, capture_error_mode="thread_local")
                try:
                    self.losses = stepper.step(*self.static, eps=self.static_eps)
                finally:
                    graphs[-1].capture_end()
        finally:
            stepper._reduce_hook = None
        torch.cuda.current_stream().wait_stream(side)
        self.segments = (graphs, flats)

    def step(self, disc_edge=None, disc_node=None, gen_edge=None, gen_node=None, check_one_hot: bool = True, eps=None):
        """Replay on a new batch (``None`` keeps the previous tensor).  Edge batches captured as one-hot must be one-hot
        again: labels attached by ``data.load_molecules`` are trusted, other tensors are validated by ``as_one_hot``
        (one device->host read per new tensor object; ``check_one_hot=False`` skips it and trusts ``argmax``)."""
        # validate every new batch BEFORE touching a static buffer: a batch that is rejected leaves the dense buffers and
        # the label buffers of the previous batch intact (and consistent with each other)
        todo = []
        for idx, (dst, src) in enumerate(zip(self.static, (disc_edge, disc_node, gen_edge, gen_node))):
            if src is None or src.data_ptr() == dst.data_ptr():
                continue
            lab = None
            if idx in self._labels:
                lab = one_hot_labels(as_one_hot(src)) if check_one_hot else one_hot_labels(src)
                if lab is None:
                    if check_one_hot:
                        raise RuntimeError("GraphedGANStep was captured with a one-hot edge batch (table-gather embedding); "
                                           "the new batch is not one-hot: capture a new graph for dense inputs")
                    lab = src.argmax(-1)
            todo.append((idx, dst, src, lab))
        if eps is not None:
            if self.static_eps is None:
                raise RuntimeError("GraphedGANStep was captured with eps drawn inside the graph: pass eps=(...) to its constructor "
                                   "to replay on given interpolation weights")
            for dst, src in zip(self.static_eps, eps):
                dst.copy_(src)
        for idx, dst, src, lab in todo:
            dst.copy_(src)
            if lab is not None:
                self._labels[idx].copy_(lab)      # in place: the captured kernels read this buffer
                attach_one_hot_labels(dst, self._labels[idx])
        if self.segments is None:
            self.graph.replay()
        else:
            graphs, flats = self.segments
            reduce = dist.is_available() and dist.is_initialized()
            for g, flat in zip(graphs, flats):
                g.replay()
                if reduce:
                    self.stepper._reduce_flat(flat, self.world)      # eager: D's bucket after the first graph, G's after the second
            graphs[-1].replay()
        # the replay updated G and D in place without bumping tensor versions: an eager forward after
        # it must not reuse packs keyed on the old versions
        bump_weights_epoch()
        return self.losses
