"""``MSDeformAttn`` — multi-scale deformable attention module, drop-in for UVHand
``models/ops/modules/ms_deform_attn.py:30-140`` (imported by
``models/arctic_transformer.py:20``, ``models/origin_deformable_transformer.py:20``,
``models/assembly_transformer.py:20``).

Contract kept: constructor signature and defaults; the attributes
``im2col_step, d_model, n_levels, n_heads, n_points``; the four ``nn.Linear``
sub-modules named ``sampling_offsets, attention_weights, value_proj, output_proj``
(checkpoint keys, and ``lr_linear_proj_names`` in util/settings.py:75, depend on the
names), created in that order so that a seeded construction consumes the RNG stream
exactly like the reference; ``_reset_parameters()`` (:62-78; called again from
outside, models/arctic_transformer.py:80-81); the ``forward`` signature and its
reference-point branches: 2-d (:105-108), the ARCTIC 42-d keypoint branch (21 (x, y)
keypoints averaged per level, :110-128) and the upstream 4-d box branch that the
dn_dab copy carries (models/dn_dab_dino_deformable_detr/ops/modules/ms_deform_attn.py:106-108);
any other width raises ``ValueError`` (:134-136).

The sampling itself is ``MSDeformAttnFunction`` (HIP kernels); the four projections
stay ``nn.Linear`` (hipBLASLt / MFMA on ROCm), as north_star prescribes.
"""
import math
import warnings
import weakref

import torch
import torch.nn.functional as F
from torch import nn

from .. import _ext, _native
from ..functions import (MSDeformAttnBF16Function, MSDeformAttnFunction, MSDeformAttnMergedPrologueFunction,
                         MSDeformAttnPrologueFunction)
from ..functions.linear_func import _autocast_dtype, bracket_linear, bracket_linear_masked, bracket_linear_wb


# The reference asserts sum_l H_l*W_l == Len_in on every call of every layer (:93) — a device->host sync
# 12x per training step.  Here each distinct spatial_shapes tensor OBJECT is read back once (the models hand
# the same tensor to all their layers, models/arctic_transformer.py:176-179,293,382); the entry is keyed on the
# object's identity and dropped when the tensor dies, so a recycled address or id can never vouch for other
# contents.  Tensors that track in-place versions are re-checked when modified; inference tensors (no version
# counter) are keyed on identity alone.  During HIP-graph capture nothing may synchronise: the check is skipped
# there — the kernels themselves never touch memory outside [0, S) whatever the shapes say (include/msda.h).
_verified_shapes = {}


def _shapes_version(t):
    try:
        return t._version
    except RuntimeError:                      # "Inference tensors do not track version counter"
        return None


def _check_shapes_sum(spatial_shapes, len_in):
    key = id(spatial_shapes)
    entry = _verified_shapes.get(key)
    want = (_shapes_version(spatial_shapes), tuple(spatial_shapes.shape), int(len_in))
    if entry is not None and entry[0]() is spatial_shapes and entry[1] == want:
        return
    if spatial_shapes.is_cuda and torch.cuda.is_current_stream_capturing():
        return
    assert (spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum() == len_in
    try:
        ref = weakref.ref(spatial_shapes, lambda _r, k=key: _verified_shapes.pop(k, None))
    except TypeError:                         # not weak-referenceable: just do not cache
        return
    _verified_shapes[key] = (ref, want)


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return n != 0 and (n & (n - 1)) == 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a per-head dimension that is a power of 2 (32 in the UVHand models) "
                          "takes the tiled HIP kernels; other widths use the generic ones.")

        self.im2col_step = 64
        # opt-in, not in the reference: keep value / sampled output in bfloat16 (fp32 accumulation)
        self.bf16_storage = False
        # softmax + location arithmetic inside the kernels where the geometry allows (fp32, 2-d / 42-d
        # reference points); False = compose them in PyTorch exactly like the reference
        self.fused_prologue = True
        # with the fused prologue: sampling_offsets and attention_weights (two layers on the same query) as
        # ONE GEMM whose output the kernels read in place; the parameters stay two nn.Linear modules
        self.merged_projection = True
        # the fp32 fused path as ONE C++ autograd node (csrc/torch_ext/msda_torch.cpp: module_forward) when the torch
        # extension is built: the same kernels queued without Python in between (the eager step is host-bound at decoder
        # sizes); False = the Python composition below
        self.cpp_node = True
        # the one-node path reads [sampling_offsets ; attention_weights] as ONE weight / bias: the four parameters are VIEWS of
        # two persistent buffers (_merge_projection_storage), so nothing is concatenated per call and nothing can go stale —
        # optimizers, load_state_dict and DDP write through the parameters into the very storage the GEMM reads.  False: the
        # node concatenates per call (two small copies), as during a stream capture that finds the views broken
        self.share_projection_storage = True
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)

        self._reset_parameters()

    def _reset_parameters(self):
        # sampling offsets start as the n_heads unit directions of a regular polygon, scaled to the
        # unit square's border and by the point index 1..n_points; attention logits start at zero.
        nn.init.constant_(self.sampling_offsets.weight.data, 0.0)
        angles = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        dirs = torch.stack([angles.cos(), angles.sin()], -1)
        dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]
        bias = dirs.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        bias = bias * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, self.n_points, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(bias.reshape(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.0)
        nn.init.constant_(self.attention_weights.bias.data, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.0)
        self._merge_projection_storage()                         # (the bias above is a new Parameter: new storage)

    def _cpp_node(self, query, reference_points, input_flatten, spatial_shapes, level_start_index, padding_mask):
        """The torch extension, if this call can run as one of its one-node forms of the fused path (module_forward /
        module_forward_bf16 in csrc/torch_ext/msda_torch.cpp): float32 CUDA tensors — outside autocast, or under
        autocast(bfloat16) with bf16_storage — 2-d / 42-d reference points, plain nn.Linear projections with biases, a
        geometry the fused prologue takes.  None -> the Python composition below."""
        if not (self.cpp_node and self.fused_prologue and self.merged_projection):
            return None
        ext = _ext.get()
        if ext is None or not hasattr(ext, "module_forward") or _native._forced_path != -1:
            return None
        if reference_points.shape[-1] not in (2, 42):               # (under no_grad the node runs its forward only)
            return None
        # float32 outside autocast -> module_forward; bf16 rows under autocast(bfloat16) -> module_forward_bf16 (the two
        # configurations the training loop runs); bf16 rows without autocast, other autocast types: the composition
        autocast = torch.is_autocast_enabled()
        if autocast != bool(self.bf16_storage):
            return None
        if autocast and not (hasattr(ext, "module_forward_bf16") and _autocast_dtype() == torch.bfloat16
                             and self.d_model % 8 == 0):
            return None
        tensors = (query, reference_points, input_flatten)
        layers = (self.sampling_offsets, self.attention_weights, self.value_proj, self.output_proj)
        if not all(t.is_cuda and t.dtype == torch.float32 for t in tensors):
            return None
        if not all(type(m) is nn.Linear and m.bias is not None and m.weight.dtype == torch.float32 and m.weight.is_cuda
                   for m in layers):
            return None
        if not (spatial_shapes.is_cuda and level_start_index.is_cuda and self.d_model % 4 == 0
                and (self.n_heads * self.n_levels * self.n_points) % 2 == 0):
            return None
        if padding_mask is not None and not (padding_mask.dtype == torch.bool and padding_mask.is_cuda
                                             and padding_mask.shape == input_flatten.shape[:-1]):
            return None
        # the node indexes device memory with these sizes (msda_torch.cpp: check_module_args raises on a mismatch): anything
        # unusual — broadcastable reference points, fewer shape rows than n_levels, replaced projection layers of another
        # width — goes to the composition below, which raises the reference's errors or broadcasts like the reference
        if query.dim() != 3 or input_flatten.dim() != 3:
            return None
        N, Len_q, C = query.shape
        mlp = self.n_heads * self.n_levels * self.n_points
        if not (C == self.d_model and input_flatten.shape[0] == N and input_flatten.shape[2] == C
                and tuple(reference_points.shape[:3]) == (N, Len_q, self.n_levels)
                and tuple(spatial_shapes.shape) == (self.n_levels, 2) and tuple(level_start_index.shape) == (self.n_levels,)
                and spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64
                and tuple(self.sampling_offsets.weight.shape) == (2 * mlp, C)
                and tuple(self.attention_weights.weight.shape) == (mlp, C)
                and tuple(self.value_proj.weight.shape) == (C, C) and tuple(self.output_proj.weight.shape) == (C, C)):
            return None
        if not _native.prologue_geometry_supported(N, input_flatten.shape[1], self.n_heads, self.d_model // self.n_heads,
                                                   self.n_levels, Len_q, self.n_points):
            return None
        return ext

    def _projection_params(self):
        return (self.sampling_offsets.weight, self.attention_weights.weight, self.sampling_offsets.bias, self.attention_weights.bias)

    def _merge_projection_storage(self):
        """Re-seat sampling_offsets.{weight,bias} and attention_weights.{weight,bias} as views of ONE [3*M*L*P, C] weight
        buffer and ONE [3*M*L*P] bias buffer (values kept).  The Parameter objects, their names and state_dict keys are
        untouched (optimizer state, lr groups by name — util/settings.py:75 — and checkpoints see two nn.Linear as before);
        only their storage is shared, so the merged GEMM of the one-node path reads the live parameters with no copy.
        Returns False (nothing changed) when the layers are not two plain float32 nn.Linear with biases on one device."""
        self.__dict__.pop("_merged", None)
        ps = self._projection_params() if (type(self.sampling_offsets) is nn.Linear and type(self.attention_weights) is nn.Linear) else (None,)
        if any(p is None for p in ps) or len({(p.dtype, p.device) for p in ps}) != 1 or ps[0].dtype != torch.float32:
            return False
        if ps[0].dim() != 2 or ps[1].dim() != 2 or ps[0].shape[1] != ps[1].shape[1]:
            return False
        n_off = ps[0].shape[0]
        with torch.no_grad():
            wbuf = torch.cat([ps[0].detach(), ps[1].detach()], 0)
            bbuf = torch.cat([ps[2].detach(), ps[3].detach()], 0)
            ps[0].data, ps[1].data = wbuf[:n_off], wbuf[n_off:]
            ps[2].data, ps[3].data = bbuf[:n_off], bbuf[n_off:]
        self.__dict__["_merged"] = (wbuf, bbuf)
        return True

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() give every parameter storage of its own: share it again
        out = super()._apply(fn, *args, **kwargs)
        if self.__dict__.get("share_projection_storage", False):
            self._merge_projection_storage()
        return out

    def _merged_projection_weights(self):
        """The [sampling_offsets ; attention_weights] weight and bias the one-node path's single GEMM reads: the buffers the
        four parameters are views of.  If something has re-seated a parameter since (load_state_dict(assign=True), a new
        nn.Parameter, `.data = ...`), the storage is shared again first; during a stream capture (no allocation may end up in
        the graph's private pool) or with share_projection_storage off: (None, None) — the node concatenates per call."""
        if not self.share_projection_storage:
            return None, None
        ps = self._projection_params()
        merged = self.__dict__.get("_merged")
        for _ in range(2):
            if merged is not None:
                wbuf, bbuf = merged
                n_off, c = ps[0].shape[0] if ps[0] is not None else -1, wbuf.shape[1]
                if (all(p is not None and p.dtype == wbuf.dtype and p.device == wbuf.device and p.is_contiguous() for p in ps)
                        and ps[0].data_ptr() == wbuf.data_ptr() and ps[1].data_ptr() == wbuf.data_ptr() + 4 * n_off * c
                        and ps[2].data_ptr() == bbuf.data_ptr() and ps[3].data_ptr() == bbuf.data_ptr() + 4 * n_off
                        and ps[0].shape[0] + ps[1].shape[0] == wbuf.shape[0] and ps[0].shape[1] == c == ps[1].shape[1]):
                    return wbuf, bbuf
            capturing = ps[0] is not None and ps[0].is_cuda and torch.cuda.is_current_stream_capturing()
            if capturing or not self._merge_projection_storage():
                return None, None
            merged = self.__dict__.get("_merged")
        return None, None

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """
        query                    (N, Len_q, C)
        reference_points         (N, Len_q, n_levels, 2) in [0, 1] (top-left (0,0), bottom-right (1,1)), or
                                 (N, Len_q, n_levels, 42): 21 (x, y) keypoints, averaged, or
                                 (N, Len_q, n_levels, 4): (cx, cy, w, h) reference boxes
        input_flatten            (N, sum_l H_l*W_l, C)
        input_spatial_shapes     (n_levels, 2) int64 [(H_0, W_0), ...]
        input_level_start_index  (n_levels,) int64 [0, H_0*W_0, ...]
        input_padding_mask       (N, sum_l H_l*W_l) bool, True at padding
        returns                  (N, Len_q, C)
        """
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        _check_shapes_sum(input_spatial_shapes, Len_in)
        # reference points given once for all levels ([N, Len_q, 1, .]): the reference's arithmetic broadcasts them (:110-128);
        # the fused paths index a row per level, so make the broadcast explicit (a view)
        if reference_points.dim() == 4 and reference_points.shape[2] == 1 and self.n_levels > 1:
            reference_points = reference_points.expand(-1, -1, self.n_levels, -1)

        ext = self._cpp_node(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                             input_padding_mask)
        if ext is not None:
            centre = reference_points if reference_points.shape[-1] == 2 else torch.stack(
                [reference_points[..., 0::2].mean(-1), reference_points[..., 1::2].mean(-1)], -1)
            node = ext.module_forward_bf16 if self.bf16_storage else ext.module_forward
            wm, bm = self._merged_projection_weights()
            return node(
                query, centre, input_flatten, input_padding_mask, input_spatial_shapes, input_level_start_index,
                self.sampling_offsets.weight, self.sampling_offsets.bias, self.attention_weights.weight,
                self.attention_weights.bias, self.value_proj.weight, self.value_proj.bias, self.output_proj.weight,
                self.output_proj.bias, self.n_heads, self.n_levels, self.n_points, self.im2col_step,
                _native.deterministic_requested(), wm, bm)

        # the four projections are nn.Linear (same parameters, same forward GEMM as the reference);
        # bracket_linear only swaps the weight-gradient GEMM of their backward (functions/linear_func.py)
        if input_padding_mask is not None:
            value = bracket_linear_masked(input_flatten, self.value_proj, input_padding_mask)
        else:
            value = bracket_linear(input_flatten, self.value_proj)
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        ref_dim = reference_points.shape[-1]

        if (self.fused_prologue and self.merged_projection and ref_dim in (2, 42) and value.is_cuda
                # fp32 throughout, or bf16 rows (then also under autocast: the query projection is pinned to fp32 below)
                and ((value.dtype == torch.float32 and query.dtype == torch.float32
                      and reference_points.dtype == torch.float32 and not torch.is_autocast_enabled())
                     or (self.bf16_storage and value.dtype in (torch.float32, torch.bfloat16)))
                # the kernels move (x, y) pairs as 8 bytes: the offsets block starts every 3*M*L*P floats
                and (self.n_heads * self.n_levels * self.n_points) % 2 == 0
                and _native.prologue_geometry_supported(N, Len_in, self.n_heads, self.d_model // self.n_heads,
                                                        self.n_levels, Len_q, self.n_points)):
            autocast = torch.is_autocast_enabled()
            with torch.autocast("cuda", enabled=False):
                # sampling offsets, logits and reference points stay float32 whatever the rows are stored as: a bf16
                # location has a resolution of 0.2 pixel on a 48-pixel map
                ref32 = reference_points.float()
                centre = ref32 if ref_dim == 2 else torch.stack([ref32[..., 0::2].mean(-1), ref32[..., 1::2].mean(-1)], -1)
                projected = bracket_linear_wb(
                    query.float(), torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0).float(),
                    torch.cat([self.sampling_offsets.bias, self.attention_weights.bias], 0).float())
                output = MSDeformAttnMergedPrologueFunction.apply(
                    value, input_spatial_shapes, input_level_start_index, centre, projected, self.im2col_step,
                    self.n_heads, self.n_levels, self.n_points, self.bf16_storage)
            if self.bf16_storage and not autocast:
                output = output.to(self.output_proj.weight.dtype)
            return bracket_linear(output, self.output_proj)

        sampling_offsets = bracket_linear(query, self.sampling_offsets).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        attention_weights = bracket_linear(query, self.attention_weights).view(
            N, Len_q, self.n_heads, self.n_levels * self.n_points)

        if self.fused_prologue and not self.bf16_storage and ref_dim in (2, 42):
            # reference point per level: given (2-d) or the mean of the 21 keypoints (42-d, :121-122)
            centre = reference_points if ref_dim == 2 else torch.stack(
                [reference_points[..., 0::2].mean(-1), reference_points[..., 1::2].mean(-1)], -1)
            if _native.prologue_supported(value, centre, sampling_offsets, attention_weights):
                output = MSDeformAttnPrologueFunction.apply(
                    value, input_spatial_shapes, input_level_start_index, centre, sampling_offsets,
                    attention_weights, self.im2col_step)
                return bracket_linear(output, self.output_proj)

        attention_weights = F.softmax(attention_weights, -1).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points)

        if ref_dim == 2 or ref_dim == 42:
            # offsets are in pixels of each level: normalise by (W_l, H_l)
            wh = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            if ref_dim == 2:
                centre = reference_points[:, :, None, :, None, :]
            else:
                cx = reference_points[:, :, None, :, None, 0::2].mean(-1).unsqueeze(-1)
                cy = reference_points[:, :, None, :, None, 1::2].mean(-1).unsqueeze(-1)
                centre = torch.cat([cx, cy], dim=-1)
            sampling_locations = centre + sampling_offsets / wh[None, None, None, :, None, :]
        elif ref_dim == 4:
            sampling_locations = reference_points[:, :, None, :, None, :2] \
                + sampling_offsets / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(
                "Last dim of reference_points must be 2 or 4, but get {} instead.".format(ref_dim))

        fn = MSDeformAttnBF16Function if self.bf16_storage else MSDeformAttnFunction
        if value.dtype == torch.float16 and not self.bf16_storage:
            # all-half inputs (amp): float32 inside the op, half outside — the dino copy of the module,
            # models/dino/ops/modules/ms_deform_attn.py:124-131
            output = fn.apply(value.float(), input_spatial_shapes, input_level_start_index, sampling_locations.float(),
                              attention_weights.float(), self.im2col_step).to(torch.float16)
            return bracket_linear(output, self.output_proj)
        output = fn.apply(value, input_spatial_shapes, input_level_start_index, sampling_locations,
                          attention_weights, self.im2col_step)
        return bracket_linear(output.to(self.output_proj.weight.dtype) if self.bf16_storage else output,
                              self.output_proj)
