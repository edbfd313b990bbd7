"""Twice-differentiable autograd wrappers around the HIP kernels.

Each op is a pair of ``torch.autograd.Function``s: the forward op, and its
backward expressed as a second Function whose own backward calls the
second-order kernel.  That keeps the reference's gradient penalty
(``src/model/loss.py:32-39``: ``autograd.grad(..., create_graph=True)`` followed
by ``d_loss.backward()``, ``train.py:367``) working unchanged on these modules.

One module per autograd node family, layered (each imports the ones before it; this package re-exports all of them, private
helpers included, so ``druggen_amd.functional.<name>`` resolves whatever module ``<name>`` lives in):

    _runtime    shared state: traffic accounting, hidden-tensor storage modes, workspaces, activation dtype, pass flags,
                reduce-batch / riding-launch scopes, the packed-weight caches' epoch
    layernorm   residual + LayerNorm and its two backward orders
    dense       nn.Linear: weight gradients, packed weights, row GEMMs with fused prologue / epilogue, q / k / v per launch
    heads       readouts, node embedding chain, Discriminator head
    ffn         feed-forward half of an Encoder_Block (float32 fused forward / two launches, bf16 fused)
    attention   attention core and the fused attention half (float32, bf16)
    embed       edge embedding + symmetrisation, one-hot table form, output slots
"""
from __future__ import annotations

from . import _runtime, layernorm, dense, heads, ffn, attention, embed      # noqa: F401
from ._runtime import *      # noqa: F401,F403
from .layernorm import *     # noqa: F401,F403
from .dense import *       # noqa: F401,F403
from .heads import *         # noqa: F401,F403
from .ffn import *           # noqa: F401,F403
from .attention import *     # noqa: F401,F403
from .embed import *         # noqa: F401,F403

__all__ = ["attach_one_hot_labels", "attn_core", "ln_residual", "linear", "linear_relu", "linear_ln", "ffn_ln", "attn_block", "embed_sym", "inputs_only_backward",
           "second_order_forward", "in_second_order_forward", "readout", "traffic_reset", "traffic_bytes", "traffic_flops", "traffic_floor_bytes",
           "set_activation_dtype", "activation_dtype", "hidden_storage", "hidden_forward_storage", "hidden_to_float", "activations", "as_one_hot", "one_hot_labels", "embed_sym_onehot", "OutSlot", "join_parts", "set_fused_ffn_f32"]
