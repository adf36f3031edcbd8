"""``add_layer_norm(x, residual, norm)`` — ``norm(x + residual)`` for an ``nn.LayerNorm`` over the last dimension as
ONE kernel per direction (``msda_add_layernorm_*_f32``, include/msda.h) instead of an add and a LayerNorm
(UVHand models/arctic_transformer.py:279-282, 294-295, 366-368, 377-378, 385-386: ``x = x + dropout(x2); x = norm(x)``).

The dropout stays the caller's (PyTorch's own ``nn.Dropout`` on ``residual``), so training keeps the framework's random
stream.  Falls back to ``norm(x + residual)`` — the framework's implementation of the same two layers — when the
kernel's preconditions do not hold (CPU tensors, non-fp32 / autocast, widths beyond 1024 or not a multiple of 4,
LayerNorm without affine parameters), so the layers work wherever the reference's do."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _native as MSDA


class _AddLayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps):
        y, mean, rstd = MSDA.add_layernorm_forward(x, residual, weight, bias, eps)
        ctx.save_for_backward(x, residual, weight, mean, rstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        x, residual, weight, mean, rstd = ctx.saved_tensors
        gs, gw, gb = MSDA.add_layernorm_backward(grad_y.contiguous(), x, residual, weight, mean, rstd)
        need = ctx.needs_input_grad
        return (gs if need[0] else None, gs if need[1] else None, gw if need[2] else None, gb if need[3] else None, None)


def add_layer_norm(x, residual, norm):
    """``norm(x + residual)`` (``residual`` may be None: plain ``norm(x)``)."""
    if (norm.elementwise_affine and norm.bias is not None and len(norm.normalized_shape) == 1
            and not torch.is_autocast_enabled() and x.is_cuda and x.dtype == torch.float32):
        xc = x.contiguous()
        rc = residual.contiguous() if residual is not None else None
        if MSDA.add_layernorm_supported(xc, rc, norm.weight, norm.bias):
            return _AddLayerNormFn.apply(xc, rc, norm.weight, norm.bias, norm.eps)
    return norm(x if residual is None else x + residual)
