"""Decoder self-attention of the drop-in layers (UVHand models/arctic_transformer.py:351, :374-376:
``nn.MultiheadAttention(d_model, n_heads, dropout)`` called sequence-first with q = k = tgt + query_pos, v = tgt, no masks) with
the attention core — ``dropout(softmax(q k^T / sqrt(d))) v`` — on the library's own kernels (msda_attn32_*_f32, include/msda.h)
when it fits them: head_dim 32, at most 320 queries, fp32, no masks.  The module, its parameters and state_dict keys stay
``nn.MultiheadAttention``'s; the in- and out-projections are ``nn.Linear``'s arithmetic on the module's own weights (q and k share
their input, so their two projections are ONE GEMM on the first two thirds of ``in_proj_weight``), with the weight gradients on the
library's MFMA kernel like every other projection of the drop-in layers (``functions/linear_func.py``).

What differs from the stock module: no [N*heads, L, L] tensor is ever written (scores, probabilities, dropout mask), and the
attention dropout draws its mask from the kernel's own hash of (seed, head, query, key) — the seed comes from torch's generator
(one ``random_()`` on a device scalar: reproducible under ``manual_seed``, capturable in a HIP graph), the stream is not
``nn.functional.dropout``'s.  Anything the kernels do not take goes to the module itself."""
import math

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _native as MSDA
from .linear_func import bracket_linear_wb


class _Attn32Fn(Function):
    """core(qk [L, N, 2E] = packed q | k projections, v [L, N, E]) -> [L, N, E]"""

    @staticmethod
    def forward(ctx, qk, v, heads, dropout_p):
        E = v.shape[2]
        q, k = qk[..., :E], qk[..., E:]
        scale = 1.0 / math.sqrt(E // heads)
        seed = torch.empty((), dtype=torch.int64, device=v.device).random_().view(1) if dropout_p > 0 else None
        out, lse = MSDA.attn32_forward(q, k, v, heads, scale, dropout_p, seed)
        ctx.save_for_backward(qk, v, out, lse, *([seed] if seed is not None else []))
        ctx.heads, ctx.scale, ctx.dropout_p = heads, scale, dropout_p
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        qk, v, out, lse = ctx.saved_tensors[:4]
        seed = ctx.saved_tensors[4] if len(ctx.saved_tensors) > 4 else None
        E = v.shape[2]
        g_qk = torch.empty_like(qk, memory_format=torch.contiguous_format)
        _, _, g_v = MSDA.attn32_backward(qk[..., :E], qk[..., E:], v, out, lse, grad_out, ctx.heads, ctx.scale, ctx.dropout_p, seed,
                                         grad_q=g_qk[..., :E], grad_k=g_qk[..., E:])
        return g_qk, g_v, None, None


def _takes(mha, x_qk, x_v):
    E = mha.embed_dim
    return (type(mha) is nn.MultiheadAttention and mha._qkv_same_embed_dim and not mha.batch_first and mha.bias_k is None
            and mha.bias_v is None and not mha.add_zero_attn and mha.in_proj_bias is not None and mha.head_dim == 32
            and x_qk.is_cuda and x_qk.dtype == torch.float32 and x_v.dtype == torch.float32 and x_qk.dim() == 3
            and x_qk.shape == x_v.shape and x_qk.shape[2] == E and not torch.is_autocast_enabled()
            and MSDA.attn32_supported(x_qk.shape[0], x_v.shape[0], mha.head_dim))


def self_attention(mha, x_qk, x_v):
    """``mha(x_qk, x_qk, x_v, need_weights=False)[0]`` for sequence-first inputs [L, N, E]."""
    if not _takes(mha, x_qk, x_v):
        return mha(x_qk, x_qk, x_v, need_weights=False)[0]
    E = mha.embed_dim
    w, b = mha.in_proj_weight, mha.in_proj_bias
    # (the three projections through bracket_linear_wb: nn.Linear's forward, the weight gradients on the library's split-M MFMA
    # kernel — torch runs the 9600-row ones at a third of its rate, 61 against 26 us each, tools/wgrad_time.py)
    qk = bracket_linear_wb(x_qk, w[:2 * E], b[:2 * E])
    v = bracket_linear_wb(x_v, w[2 * E:], b[2 * E:])
    core = _Attn32Fn.apply(qk, v, mha.num_heads, float(mha.dropout) if mha.training else 0.0)
    return bracket_linear_wb(core, mha.out_proj.weight, mha.out_proj.bias)
