"""Run-time options of druggen_amd: the whole switchboard in one place.

Environment variables are read ONCE, at import, and only the documented ones below (README "Switches"); everything else about a
launch is decided by shapes and dtypes.  Programmatic access: ``options.set(name=value)`` / ``with options.override(name=value)``.
The C library (libdruggen_hip.so) reads no environment variable at all.

    variable                values (first = default)        what it selects
    DG_HIDDEN               dh16 | dh24 | f32 | f24 | f16   storage of the float32 feed-forward's [R,384] hidden tensors (functional.hidden_storage)
    DG_FFN_F32              fused | unfused                 float32 feed-forward forward as ONE kernel (hidden tensor on chip) or two row GEMMs
    DG_ATTN_HALF_F32        fused | n48 | off               float32 fused attention-half forward: N <= 96 | N <= 48 only | three launches
    DG_ATTN_HALF_F32_BWD    fused | nograph | off | force   float32 fused attention-half backward part 1 (force: any batch size, tests)
    DG_ATTN_HALF            fused | unfused | force         bf16 fused attention half (force: also 48 < N <= 96)
    DG_FFN_BF16             fused | unfused                 bf16 feed-forward as fused kernels or row GEMMs
    DG_LOW_MEMORY_SHARE     auto | on | off                 low-memory step: keep the generator's graph through the D step (trainer.GANStep)
    DG_LIB                  <path>                          another build of libdruggen_hip.so (developer A/B builds; read by _lib.py)
    DG_DIST_BACKEND         nccl | gloo                     bench.py's process-group backend (gloo: functional test of N > 1 on one GPU)
    DG_FORCE_REBUILD        0 | 1                           __graft_entry__.build(): recompile every translation unit

Attributes WITHOUT an environment variable are hooks of the equivalence tests (tests/: fused launch == the launches it replaces);
their defaults are what ships.
"""
from __future__ import annotations

import contextlib
import os

_CHOICES = {
    "hidden": ("dh16", "dh24", "f32", "f24", "f16"),
    "ffn_f32": ("fused", "unfused"),
    "attn_half_f32": ("fused", "n48", "off"),
    "attn_half_f32_bwd": ("fused", "nograph", "off", "force"),
    "attn_half": ("fused", "unfused", "force"),
    "ffn_bf16": ("fused", "unfused"),
    "low_memory_share": ("auto", "on", "off"),
    # equivalence-test hooks (no environment variable)
    "ln_bwd_epilogue": (True, False),      # ln6's backward in the epilogue of the next block's dy GEMM (dg_row_gemm_ln_bwd)
    "ln_bwd_prologue": (True, False),      # ln4's backward in the producers of the out_e input-gradient GEMM (dg_row_gemm_ln_bwd_in)
    "ffn_pair": (True, False),             # node + edge feed-forward of a block as one autograd node (riding launches)
    "penalty_wgrad": ("joined", "engine"), # the penalty's second-order parameter gradients joined in the forward node
    "embed_bf16": ("fast", "general"),     # bf16 edge-embedding backward: streaming kernel or the general one
}
_ENV = {"hidden": "DG_HIDDEN", "ffn_f32": "DG_FFN_F32", "attn_half_f32": "DG_ATTN_HALF_F32",
        "attn_half_f32_bwd": "DG_ATTN_HALF_F32_BWD", "attn_half": "DG_ATTN_HALF", "ffn_bf16": "DG_FFN_BF16",
        "low_memory_share": "DG_LOW_MEMORY_SHARE"}


class _Options:
    def __init__(self):
        for name, choices in _CHOICES.items():
            value = choices[0]
            env = _ENV.get(name)
            if env and os.environ.get(env) is not None:
                value = os.environ[env]
                if value not in choices:
                    raise ValueError(f"{env}={value!r}: expected one of {choices}")
            object.__setattr__(self, name, value)

    def __setattr__(self, name, value):
        if name not in _CHOICES:
            raise AttributeError(f"druggen_amd.options has no option {name!r}")
        if value not in _CHOICES[name]:
            raise ValueError(f"options.{name} = {value!r}: expected one of {_CHOICES[name]}")
        object.__setattr__(self, name, value)

    def set(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @contextlib.contextmanager
    def override(self, **kw):
        prev = {k: getattr(self, k) for k in kw}
        self.set(**kw)
        try:
            yield self
        finally:
            self.set(**prev)

    def as_dict(self):
        return {k: getattr(self, k) for k in _CHOICES}


options = _Options()
