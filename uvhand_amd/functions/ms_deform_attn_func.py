"""``MSDeformAttnFunction`` — the autograd entry point of the op, API-compatible with
UVHand ``models/ops/functions/ms_deform_attn_func.py:21-39`` but backed by the HIP
kernels in ``uvhand_amd/csrc`` through the C ABI of ``include/msda.h``.

    out = MSDeformAttnFunction.apply(value[N,S,M,D], value_spatial_shapes int64[L,2],
                                     value_level_start_index int64[L],
                                     sampling_locations[N,Lq,M,L,P,2],
                                     attention_weights[N,Lq,M,L,P], im2col_step)
    -> out[N, Lq, M*D];  grads for value, sampling_locations, attention_weights only.

Behaviour kept from the reference: ``value`` is cast to the compute dtype before the
native call in forward and in backward (:26,:37 cast it to float32; here the compute
dtype is that of ``sampling_locations`` — float32 in the models, float64 under the
reference test's gradcheck, models/ops/test.py:76, which the upstream un-cast
wrapper supports); the tensors saved for backward are the un-cast inputs (:28);
backward is once-differentiable (:32); CPU tensors raise (src/ms_deform_attn.h:38).
There is no PyTorch fallback in this package: the reference's debug helper
``ms_deform_attn_core_pytorch`` (:42-62) is restated only under ``oracle/`` as test
infrastructure.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _ext
from .. import _native as MSDA

_EXT_DTYPES = (torch.float32, torch.float64)


class MSDeformAttnFunction(Function):
    @classmethod
    def apply(cls, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
              im2col_step):
        """Same node, two hosts: when the torch extension is built (``_ext``) and the tensors are the common case —
        CUDA, one floating dtype — the forward/backward pair below runs as a C++ autograd node that calls the same
        C ABI; otherwise (bf16/fp16 value, CPU tensors -> the reference's error, extension not built) this Python class."""
        ext = _ext.get()
        if (ext is not None and torch.is_tensor(value) and value.is_cuda and value.dtype in _EXT_DTYPES
                and torch.is_tensor(sampling_locations) and sampling_locations.dtype == value.dtype
                and torch.is_tensor(attention_weights) and attention_weights.dtype == value.dtype
                and not torch.is_autocast_enabled() and MSDA._forced_path == -1):
            return ext.apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                             int(im2col_step), MSDA.deterministic_requested())
        return super().apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                             im2col_step)

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        cdt, loc, attn = _compute_dtype(sampling_locations, attention_weights)
        # a forward whose backward will run also leaves its per-point table for it (small problems; None otherwise)
        output, table = MSDA.ms_deform_attn_forward(
            value.to(cdt), value_spatial_shapes, value_level_start_index, loc, attn, ctx.im2col_step,
            with_table=any(ctx.needs_input_grad))
        ctx.has_table = table is not None
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights, *([table] if table is not None else []))
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights = \
            ctx.saved_tensors[:5]
        table = ctx.saved_tensors[5] if ctx.has_table else None
        cdt, loc, attn = _compute_dtype(sampling_locations, attention_weights)
        # the reference asserts contiguity of grad_output (ms_deform_attn_cuda.cu:98) and would
        # raise on e.g. an expanded gradient; making it contiguous is the superset behaviour.
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value.to(cdt), value_spatial_shapes, value_level_start_index, loc, attn,
            grad_output.to(cdt).contiguous(), ctx.im2col_step, table=table)
        if loc is not sampling_locations:                                           # half inputs: gradients in their dtypes
            grad_sampling_loc = grad_sampling_loc.to(sampling_locations.dtype)
            grad_attn_weight = grad_attn_weight.to(attention_weights.dtype)
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None


def _compute_dtype(sampling_locations, attention_weights):
    """(compute dtype, locations, weights as handed to the kernels).  float32 / float64 locations select the kernel
    instantiation (the reference: AT_DISPATCH_FLOATING_TYPES, ms_deform_attn_cuda.cu:64) and are passed on untouched — a
    weights tensor of another dtype then raises as in the reference.  Half-precision locations (the amp branch of the dino
    copy of the module, models/dino/ops/modules/ms_deform_attn.py:124-131, up-casts before the call) are computed in float32."""
    if sampling_locations.dtype in _EXT_DTYPES:
        return sampling_locations.dtype, sampling_locations, attention_weights
    return torch.float32, sampling_locations.float(), attention_weights.float()


class MSDeformAttnBF16Function(Function):
    """bfloat16-storage variant — new capability, no reference counterpart (the reference op is
    fp32/fp64 only: AT_DISPATCH_FLOATING_TYPES, ms_deform_attn_cuda.cu:64,134; BASELINE config 3).

    ``value`` is rounded to bfloat16 and the output / grad_value come back in bfloat16 (half the
    HBM bytes of every row gather and store); sampling locations, attention weights and their
    gradients stay float32, and all arithmetic and accumulation inside the kernels is float32 with
    one rounding at the final store.  Same call signature as ``MSDeformAttnFunction``; D = 32 takes the tiled
    kernels, any other D the generic ones (fp32 grad_value inside, rounded here).
    """

    @classmethod
    def apply(cls, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
              im2col_step):
        ext = _ext.get()                       # the same node in C++ when the torch extension is built (see above)
        if (ext is not None and hasattr(ext, "apply_bf16") and torch.is_tensor(value) and value.is_cuda
                and value.dtype in (torch.float32, torch.bfloat16) and torch.is_tensor(sampling_locations)
                and sampling_locations.is_floating_point() and torch.is_tensor(attention_weights)
                and attention_weights.is_floating_point() and not torch.is_autocast_enabled()
                and MSDA._forced_path == -1):
            return ext.apply_bf16(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                  attention_weights, int(im2col_step), MSDA.deterministic_requested())
        return super().apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                             im2col_step)

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        output, table = MSDA.ms_deform_attn_forward(
            value.to(torch.bfloat16), value_spatial_shapes, value_level_start_index,
            sampling_locations.float(), attention_weights.float(), ctx.im2col_step, with_table=any(ctx.needs_input_grad))
        ctx.has_table = table is not None
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights, *([table] if table is not None else []))
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights = \
            ctx.saved_tensors[:5]
        table = ctx.saved_tensors[5] if ctx.has_table else None
        # grad_value in float32 straight from the kernel when that is what `value` needs anyway, and for long
        # backwards (several query chunks accumulate: fp32 in place, one rounding at the end)
        # — and wherever the generic kernels serve the call (D != 32: fp32 atomics are their only accumulation)
        Lq, P = sampling_locations.shape[1], sampling_locations.shape[4]
        fp32_gv = (value.dtype == torch.float32 or MSDA.backward_passes(Lq, P) > 1
                   or MSDA.path_for(2, value.shape[2], value.shape[3], value_spatial_shapes.shape[0], P) != MSDA.PATH_D32)
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value.to(torch.bfloat16), value_spatial_shapes, value_level_start_index,
            sampling_locations.float(), attention_weights.float(),
            grad_output.to(torch.bfloat16).contiguous(), ctx.im2col_step, fp32_grad_value=fp32_gv, table=table)
        return (grad_value.to(value.dtype), None, None, grad_sampling_loc.to(sampling_locations.dtype),
                grad_attn_weight.to(attention_weights.dtype), None)


class MSDeformAttnPrologueFunction(Function):
    """The op with the module's prologue folded in (no reference counterpart as a function —
    it computes what models/ops/modules/ms_deform_attn.py:101-108 + MSDeformAttnFunction compute):

        out = apply(value[N,S,M,D], spatial_shapes, level_start_index, reference_points[N,Lq,L,2],
                    sampling_offsets[N,Lq,M,L,P,2] (pixels), attn_logits[N,Lq,M,L*P], im2col_step)

    softmax over the L*P logits and ``loc = reference_point + offset / (W_l, H_l)`` run in the forward
    kernel's point lanes; the backward kernel returns the gradients of the RAW tensors (softmax backward,
    offset scaling and the reduction over heads / points for the reference points included).  fp32 only;
    callers check ``_native.prologue_supported`` first (the module does)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, reference_points, sampling_offsets,
                attn_logits, im2col_step):
        N, Lq, M, L, P, _ = sampling_offsets.shape
        out, loc, attn, table = MSDA.ms_deform_attn_forward_prologue(
            value, value_spatial_shapes, value_level_start_index, reference_points.contiguous(),
            sampling_offsets.contiguous(), attn_logits.contiguous().view(N, Lq, M, L * P), im2col_step,
            with_table=any(ctx.needs_input_grad))
        ctx.has_table = table is not None
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, loc, attn,
                              *([table] if table is not None else []))
        ctx.logits_shape = attn_logits.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, value_spatial_shapes, value_level_start_index, loc, attn = ctx.saved_tensors[:5]
        gv, goff, glog, gref = MSDA.ms_deform_attn_backward_prologue(
            value, value_spatial_shapes, value_level_start_index, loc, attn, grad_output.contiguous(),
            table=ctx.saved_tensors[5] if ctx.has_table else None)
        return gv, None, None, gref, goff, glog.view(ctx.logits_shape), None


class MSDeformAttnMergedPrologueFunction(Function):
    """``MSDeformAttnPrologueFunction`` fed by ONE projection for offsets and logits (the
    module's ``sampling_offsets`` and ``attention_weights`` layers, models/ops/modules/ms_deform_attn.py:100-101,
    read the same ``query``):

        out = apply(value, spatial_shapes, level_start_index, reference_points[N,Lq,L,2],
                    projected[N,Lq,3*M*L*P], im2col_step, M, L, P, bf16_rows=False)

    ``projected[..., :2*M*L*P]`` are the raw offsets ([M,L,P,2] order), the rest the logits ([M,L*P]); the
    kernels read both in place (row stride 3*M*L*P) and the backward writes their gradients into one tensor
    of the same layout, so the projection's backward is one input-gradient GEMM and one weight-gradient GEMM
    instead of two of each plus an add.

    ``bf16_rows`` (BASELINE config 3; no reference counterpart): ``value`` is rounded to bfloat16 here (or taken as it
    is if already bfloat16, e.g. under autocast), the sampled output comes back in bfloat16, the backward reads
    bfloat16 rows and returns ``grad_value`` in float32 — accumulated in float32, rounded nowhere — cast to
    ``value``'s dtype only if that was bfloat16.  Offsets, logits, reference points and their gradients stay float32."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, reference_points, projected, im2col_step,
                n_heads, n_levels, n_points, bf16_rows=False):
        N, Lq, width = projected.shape
        mlp = n_heads * n_levels * n_points
        if width != 3 * mlp:
            raise RuntimeError("projected tensor must hold 2*M*L*P offsets and M*L*P logits per query (got %d, "
                               "expected %d)" % (width, 3 * mlp))
        projected = projected.contiguous()
        rows = value.to(torch.bfloat16) if bf16_rows else value
        out, loc, attn, table = MSDA.ms_deform_attn_forward_prologue(
            rows, value_spatial_shapes, value_level_start_index, reference_points.contiguous(),
            projected[..., :2 * mlp].view(N, Lq, n_heads, n_levels, n_points, 2),
            projected[..., 2 * mlp:].view(N, Lq, n_heads, n_levels * n_points), im2col_step,
            with_table=any(ctx.needs_input_grad))
        ctx.has_table = table is not None
        ctx.save_for_backward(rows, value_spatial_shapes, value_level_start_index, loc, attn,
                              *([table] if table is not None else []))
        ctx.value_dtype = value.dtype
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rows, value_spatial_shapes, value_level_start_index, loc, attn = ctx.saved_tensors[:5]
        gv, _, _, gref, gproj = MSDA.ms_deform_attn_backward_prologue(
            rows, value_spatial_shapes, value_level_start_index, loc, attn, grad_output.to(rows.dtype).contiguous(), merged=True,
            table=ctx.saved_tensors[5] if ctx.has_table else None)
        return gv.to(ctx.value_dtype), None, None, gref, gproj, None, None, None, None, None
