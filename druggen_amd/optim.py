"""AdamW with the reference's hyper-parameters (``train.py:213-214``:
``torch.optim.AdamW(params, lr, [beta1, beta2])``, default eps 1e-8 and weight
decay 0.01) as ONE kernel over a flat parameter buffer.

The live parameters (those that received a gradient at the first step -- the
Discriminator's dead last-block edge branch never does, and torch skips
``grad is None`` parameters entirely) are re-pointed to views of one flat fp32
buffer; gradients are packed into a matching flat buffer, which is also the
bucket the data-parallel all-reduce runs on (``trainer.GANStep``).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import _lib
from .functional import _dev, repack_params

__all__ = ["FlatAdamW"]


class FlatAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.step_count = 0
        self.device_step = None     # int32 device counter (graph-replay-safe stepping)
        self._live: Optional[List[int]] = None
        self.flat_param = self.flat_grad = self.exp_avg = self.exp_avg_sq = None
        self._grad_views: List[torch.Tensor] = []

    # -- torch.optim-like surface ------------------------------------------------
    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _build(self, live: List[int]) -> None:
        ps = [self.params[i] for i in live]
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        old_m, old_v = {}, {}
        if self._live is not None:      # liveness changed: carry the moments over
            off = 0
            for i in self._live:
                k = self.params[i].numel()
                old_m[i], old_v[i] = self.exp_avg[off:off + k].clone(), self.exp_avg_sq[off:off + k].clone()
                off += k
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.empty(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._grad_views = []
        off = 0
        with torch.no_grad():
            for i, p in zip(live, ps):
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + k].view_as(p)
                self._grad_views.append(self.flat_grad[off:off + k].view_as(p))
                if i in old_m:
                    self.exp_avg[off:off + k].copy_(old_m[i])
                    self.exp_avg_sq[off:off + k].copy_(old_v[i])
                off += k
        self._live = list(live)

    def pack_grads(self) -> Optional[torch.Tensor]:
        """Copy the live ``.grad`` tensors into the flat bucket and return it."""
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not live:
            return None
        if not self.params[live[0]].is_cuda:
            raise RuntimeError("FlatAdamW runs on the GPU (no CPU fallback)")
        if live != self._live:
            self._build(live)
        torch._foreach_copy_(self._grad_views, [self.params[i].grad for i in live])
        return self.flat_grad

    def step(self, packed: bool = False) -> None:
        if not packed and self.pack_grads() is None:
            return
        if self.flat_grad is None:
            return
        self.step_count += 1
        lib = _lib.load()
        if self.device_step is None:
            self.device_step = torch.full((1,), self.step_count - 1, dtype=torch.int32, device=self.flat_param.device)
        with _dev(self.flat_param):
            # step count in device memory: the launch stays correct when captured into a hipGraph
            _lib.check(lib.dg_adamw_flat_devstep(self.flat_param.data_ptr(), self.flat_grad.data_ptr(),
                                                 self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                                 self.flat_param.numel(), self.lr, self.betas[0], self.betas[1],
                                                 self.eps, self.weight_decay, self.device_step.data_ptr(),
                                                 _lib.stream_of(self.flat_param)), "dg_adamw_flat_devstep")
        # the kernel wrote the parameters behind autograd's back: bump their version counters so that
        # version-keyed caches (packed GEMM weights) notice, exactly as an in-place torch op would
        # (only THESE parameters go stale: a global epoch bump here made every step re-pack the other network's
        # unchanged weights too -- half of the ~230 pack launches per step)
        torch.autograd.graph.increment_version([self.params[i] for i in self._live])
        repack_params(self.params)      # their cached GEMM packs, in one launch
