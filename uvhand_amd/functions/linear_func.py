"""``bracket_linear`` — ``F.linear`` for the four projections around the sampling kernel
(UVHand models/ops/modules/ms_deform_attn.py:96,100,101,139) with the weight / bias gradient on the
hand-written split-M MFMA kernel (``msda_linear_wgrad_f32``, include/msda.h).

The forward ``x @ W^T + b`` and the input gradient ``dy @ W`` run at 38-97 TFLOP/s on the vendor BLAS at
these shapes and stay there when the layer is large; small layers take the library's own plain-launch kernels
(``_own_rows`` below: host cost).  The weight gradient ``dy^T @ x`` with few rows (600 at the
300-query decoder shape) is run by hipBLASLt as one 256x256 tile on one CU — 141 us, 0.6 TFLOP/s
(tools/gemm_baseline.py).  Numerics: fp32 MFMA is an exact fp32 fma chain; the split-M partial sums
are combined in a fixed order, so the result is reproducible run to run.

Under ``torch.autocast(bfloat16)`` the same is done on bf16 operands (``_BracketLinearAmpFn``).  Falls back to
``F.linear`` — PyTorch's own implementation of the same layer, not an alternative implementation of the op —
whenever the kernels' preconditions do not hold (CPU tensors, other dtypes, fp16 autocast, feature counts not
multiples of 4), so the module keeps working everywhere nn.Linear does.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _native as MSDA


def _own_rows(a2, weight, reduce_dim):
    """The library's plain-launch fp32-MFMA forward / input-gradient kernels (msda_linear_forward_f32 / _dgrad_f32) take the
    layer while it is small — ~7 us of host time per call against ~18-27 us for a GEMM through torch, and as fast on the
    GPU up to ~5000 rows at 256 features (tools/gemm_time.py); beyond, the vendor BLAS's large macro-tiles run 1.3x faster.
    The same rule as own_linear() in csrc/torch_ext/msda_torch.cpp, so both forms of the module run the same kernels."""
    return (a2.shape[0] * weight.shape[0] <= 4800 * 256 and weight.shape[1] <= 512
            and MSDA.linear_rows_supported(a2, weight, reduce_dim))


def _rows_forward(x, weight, bias, mask=None):
    x2 = x.reshape(-1, x.shape[-1])
    if (x2.is_contiguous() and _own_rows(x2, weight, 1) and bias is not None and bias.is_contiguous()
            and bias.data_ptr() % 16 == 0):
        return MSDA.linear_forward(x2, weight, bias, mask).view(*x.shape[:-1], weight.shape[0])
    y = F.linear(x, weight, bias)
    if mask is not None:
        MSDA.zero_masked_rows_(y.view(-1, y.shape[-1]), mask)
    return y


def _rows_dgrad(go2, weight, mask=None):
    """go2: contiguous [rows, out]; a fresh [rows, in] buffer."""
    if _own_rows(go2, weight, 0):
        return MSDA.linear_dgrad(go2, weight, mask)
    gx2 = go2 @ weight
    return MSDA.zero_masked_rows_(gx2, mask) if mask is not None else gx2


class _BracketLinearFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _rows_forward(x, weight, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad
        go2 = grad_out.reshape(-1, grad_out.shape[-1])
        grad_x = grad_w = grad_b = None
        if need_x:
            grad_x = _rows_dgrad(go2.contiguous(), weight).view_as(x)
        if need_w or (need_b and ctx.has_bias):
            x2 = x.reshape(-1, x.shape[-1])
            go2c, x2c = go2.contiguous(), x2.contiguous()
            if MSDA.linear_wgrad_supported(go2c, x2c):
                grad_w, grad_b = MSDA.linear_wgrad(go2c, x2c, want_bias=ctx.has_bias and need_b)
            else:
                grad_w = go2c.t() @ x2c
                grad_b = go2c.sum(0) if (ctx.has_bias and need_b) else None
            if not need_w:
                grad_w = None
        return grad_x, grad_w, grad_b


class _BracketLinearAmpFn(Function):
    """The same layer under ``torch.autocast(dtype=torch.bfloat16)``: the forward is what autocast would run (operands
    rounded to bf16, bf16 output); the backward takes the bf16-operand weight-gradient kernel — fp32 products and
    accumulation, an fp32 result handed to the fp32 master parameter without a bf16 rounding in between — instead of the
    vendor's small-M bf16 GEMM (49.7 us at M = 600, 2 x 118 us at M = 33 440: profiles/r02_notes.md section 7)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        xb, wb = x.to(torch.bfloat16), weight.to(torch.bfloat16)
        ctx.save_for_backward(xb, wb)
        ctx.has_bias = bias is not None
        ctx.x_dtype = x.dtype
        return F.linear(xb, wb, bias.to(torch.bfloat16) if bias is not None else None)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        xb, wb = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad
        go2 = grad_out.reshape(-1, grad_out.shape[-1]).to(torch.bfloat16)
        grad_x = grad_w = grad_b = None
        if need_x:
            grad_x = (go2 @ wb).view_as(xb).to(ctx.x_dtype)
        if need_w or (need_b and ctx.has_bias):
            x2 = xb.reshape(-1, xb.shape[-1])
            go2c, x2c = go2.contiguous(), x2.contiguous()
            if MSDA.linear_wgrad_supported(go2c, x2c):
                grad_w, grad_b = MSDA.linear_wgrad(go2c, x2c, want_bias=ctx.has_bias and need_b)
            else:
                grad_w = (go2c.t() @ x2c).float()
                grad_b = go2c.float().sum(0) if (ctx.has_bias and need_b) else None
            if not need_w:
                grad_w = None
        return grad_x, grad_w, grad_b


class _MaskedBracketLinearFn(Function):
    """``F.linear(x, weight, bias).masked_fill(row_mask[..., None], 0)`` (value_proj + padding mask,
    models/ops/modules/ms_deform_attn.py:96-98) touching only the masked rows: the forward zeroes them in the
    GEMM output in place; the backward hands the mask to the weight-gradient kernel (masked rows of grad_out
    count as zero) and zeroes the same rows of the input gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, row_mask):
        mask = row_mask.reshape(-1).contiguous()
        y = _rows_forward(x, weight, bias, mask)
        ctx.save_for_backward(x, weight, mask)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, weight, mask = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        go2 = grad_out.reshape(-1, grad_out.shape[-1]).contiguous()
        grad_x = grad_w = grad_b = None
        if need_x:
            grad_x = _rows_dgrad(go2, weight, mask).view_as(x)
        if need_w or (need_b and ctx.has_bias):
            x2 = x.reshape(-1, x.shape[-1]).contiguous()
            if MSDA.linear_wgrad_supported(go2, x2):
                grad_w, grad_b = MSDA.linear_wgrad(go2, x2, want_bias=ctx.has_bias and need_b, row_mask=mask)
            else:                                                   # same escape as _BracketLinearFn (odd view offset, huge M)
                gom = go2.masked_fill(mask[:, None], 0)
                grad_w = gom.t() @ x2
                grad_b = gom.sum(0) if (ctx.has_bias and need_b) else None
            if not need_w:
                grad_w = None
        return grad_x, grad_w, grad_b, None


class _FusedFFNFn(Function):
    """``linear2(dropout(relu(linear1(x))))`` — the FFN of the layers (models/arctic_transformer.py:283-287, :366-370) — as ONE
    autograd node:
    * forward: bias + ReLU in the epilogue of linear1's GEMM (``torch._addmm_activation``: hipBLASLt's RELU_BIAS epilogue, or
      addmm + an in-place relu where the epilogue is not offered); dropout is PyTorch's own fused kernel
      (``torch.native_dropout``: the very call ``nn.Dropout`` makes, so the Philox stream is consumed exactly as in the
      reference's loop) — its mask is NOT kept;
    * backward: the dropout and ReLU gradients are one in-place pass over linear2's input gradient
      (``grad * scale * (a > 0)`` with ``a`` = linear2's saved input; msda_relu_dropout_backward_f32), and both weight
      gradients take the split-M MFMA kernel.
    Saved: x, a = dropout(relu(h)), the two weights — not h, not relu(h), not the mask (at the training shape 137 + 137 + 34
    MB per encoder layer less).  One difference from the reference's composition: where h is NaN the reference's
    threshold_backward lets the gradient through and this node blocks it (NaN > 0 is false) — the loss is NaN either way."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p, training):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        h = torch._addmm_activation(b1, x2, w1.t())                      # relu(x2 @ w1^T + b1)
        drop = bool(training) and 0.0 < p < 1.0
        a = torch.native_dropout(h, p, True)[0] if drop else h
        y = torch.addmm(b2, a, w2.t())
        ctx.save_for_backward(x2, a, w1, w2)
        ctx.scale = 1.0 / (1.0 - p) if drop else 1.0
        ctx.x_shape = x.shape
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x2, a, w1, w2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        go2 = grad_out.reshape(-1, grad_out.shape[-1]).contiguous()
        gw2 = gb2 = gw1 = gb1 = gx = None
        if need[3] or need[4]:
            if MSDA.linear_wgrad_supported(go2, a):
                gw2, gb2 = MSDA.linear_wgrad(go2, a, want_bias=need[4])
            else:
                gw2, gb2 = go2.t() @ a, (go2.sum(0) if need[4] else None)
            if not need[3]:
                gw2 = None
        if need[0] or need[1] or need[2]:
            gh = go2 @ w2                                                  # gradient of a, then of h in place
            MSDA.relu_dropout_backward_(gh, a, ctx.scale)
            if need[1] or need[2]:
                if MSDA.linear_wgrad_supported(gh, x2):
                    gw1, gb1 = MSDA.linear_wgrad(gh, x2, want_bias=need[2])
                else:
                    gw1, gb1 = gh.t() @ x2, (gh.sum(0) if need[2] else None)
                if not need[1]:
                    gw1 = None
            if need[0]:
                gx = (gh @ w1).view(ctx.x_shape)
        return gx, gw1, gb1, gw2, gb2, None, None


_FUSED_FFN = os.environ.get("MSDA_FUSED_FFN", "1") != "0"            # A/B knob: 0 = the composition of bracket_linear / F.relu / nn.Dropout


def fused_ffn(x, linear1, activation, dropout, linear2):
    """``linear2(dropout(activation(linear1(x))))`` for two ``nn.Linear`` layers and an ``nn.Dropout``: one autograd node
    (_FusedFFNFn) when the activation is ReLU and the layers are plain float32 CUDA layers with biases outside autocast,
    else the composition of the same modules (bracket_linear keeps the weight-gradient kernel there)."""
    if (_FUSED_FFN and _ENABLED and activation is F.relu and type(linear1) is nn.Linear and type(linear2) is nn.Linear
            and type(dropout) is nn.Dropout and not dropout.inplace and dropout.p < 1.0
            and linear1.bias is not None and linear2.bias is not None and x.is_cuda and x.dtype == torch.float32
            and all(t.dtype == torch.float32 and t.is_cuda for t in (linear1.weight, linear1.bias, linear2.weight, linear2.bias))
            and not torch.is_autocast_enabled() and linear1.out_features % 4 == 0 and linear1.in_features % 4 == 0
            and linear2.out_features % 4 == 0 and x.numel() > 0):
        return _FusedFFNFn.apply(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias, float(dropout.p),
                                 dropout.training)
    return bracket_linear(dropout(activation(bracket_linear(x, linear1))), linear2)


_ENABLED = os.environ.get("MSDA_BRACKET_LINEAR", "1") != "0"       # A/B knob: 0 = always the plain layer
_MASKED_ROWS = os.environ.get("MSDA_MASKED_ROWS", "1") != "0"      # A/B knob: 0 = masked_fill after the layer


def _kernel_applies(x, weight):
    return (_ENABLED and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and not torch.is_autocast_enabled() and torch.is_grad_enabled()
            and weight.shape[1] % 4 == 0 and weight.shape[0] % 4 == 0)


def _autocast_dtype():
    try:
        return torch.get_autocast_dtype("cuda")
    except (AttributeError, TypeError):                      # older torch
        return torch.get_autocast_gpu_dtype()


def _amp_kernel_applies(x, weight):
    """bf16 autocast on CUDA around an fp32 layer (the reference trains its transformer this way when --amp is on)."""
    return (_ENABLED and x.is_cuda and torch.is_autocast_enabled() and _autocast_dtype() == torch.bfloat16
            and weight.dtype == torch.float32 and x.dtype in (torch.float32, torch.bfloat16) and torch.is_grad_enabled()
            and weight.shape[1] % 4 == 0 and weight.shape[0] % 4 == 0)


def bracket_linear_wb(x, weight, bias):
    """``F.linear(x, weight, bias)``; custom weight-gradient kernel when applicable."""
    if _kernel_applies(x, weight):
        return _BracketLinearFn.apply(x, weight, bias)
    if _amp_kernel_applies(x, weight):
        return _BracketLinearAmpFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def bracket_linear_masked(x, layer, row_mask):
    """``layer(x).masked_fill(row_mask[..., None], 0)`` for an ``nn.Linear`` ``layer`` and a bool mask over the
    leading dimensions of ``x``; the masked rows only are touched when the kernels apply."""
    if (_MASKED_ROWS and _kernel_applies(x, layer.weight) and row_mask.dtype == torch.bool and row_mask.is_cuda
            and row_mask.shape == x.shape[:-1]):
        return _MaskedBracketLinearFn.apply(x, layer.weight, layer.bias, row_mask)
    return bracket_linear(x, layer).masked_fill(row_mask[..., None], float(0))


def bracket_linear(x, layer):
    """``layer(x)`` for an ``nn.Linear`` ``layer``; custom weight-gradient kernel when applicable."""
    if _kernel_applies(x, layer.weight):
        return _BracketLinearFn.apply(x, layer.weight, layer.bias)
    if _amp_kernel_applies(x, layer.weight):
        return _BracketLinearAmpFn.apply(x, layer.weight, layer.bias)
    return layer(x)
