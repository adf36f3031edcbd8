"""Construction of the op's arguments from a feature pyramid — the caller side of the hot path
(SURVEY.md §8 a10 / f3), mirroring what UVHand's transformers do before every ``MSDeformAttn`` call:

  * ``flatten_feature_levels``    models/arctic_transformer.py:157-177 (same lines ±6 in
                                  origin_deformable_transformer.py:156-176): per level ``[N,C,H,W] -> [N,H*W,C]``,
                                  concatenation over levels, ``pos + level_embed``, int64 ``spatial_shapes`` /
                                  ``level_start_index`` ON THE DEVICE (the kernels read them there), valid ratios
  * ``get_valid_ratio``           models/arctic_transformer.py:144-151
  * ``encoder_reference_points``  models/arctic_transformer.py:310-323 (pixel-centre grid of every level,
                                  normalised by the valid extent, then scaled into every level's frame)
  * ``decoder_reference_points``  models/arctic_transformer.py:413-419 (2-d points, or the ARCTIC 42-d = 21 (x, y)
                                  keypoints, times the valid ratios of each level)

On fp32 CUDA feature maps the flatten (both tensors, all levels, the level-embedding add) is ONE tiled-transpose HIP kernel
per direction (``msda_flatten_levels_f32`` / ``msda_unflatten_levels_f32``, csrc/msda_flatten.hip); everything else — and
the flatten on other devices / dtypes — is plain PyTorch layout work, exactly the reference's composition.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _native


class _FlattenLevelsFn(Function):
    """(src_flatten, lvl_pos_embed_flatten) = flatten(srcs..., pos_embeds..., level_embed): one tiled-transpose kernel per
    direction (msda_flatten_levels_f32 / msda_unflatten_levels_f32) instead of the strided torch.cat copies and the add."""

    @staticmethod
    def forward(ctx, level_embed, *maps):
        L = len(maps) // 2
        srcs, poss = maps[:L], maps[L:]
        ctx.shapes = [tuple(t.shape) for t in srcs]
        ctx.L = L
        src_flat, pos_flat = _native.flatten_levels(srcs, poss, level_embed)
        return src_flat, pos_flat

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_src_flat, grad_pos_flat):
        L, need = ctx.L, ctx.needs_input_grad
        need_src, need_pos = any(need[1:1 + L]), any(need[1 + L:])
        gs = gp = g_embed = None
        if need_src or need_pos or need[0]:
            # the level-embedding gradient (column sums of grad_pos_flat per level) rides along with the positional copy
            gs, gp, g_embed = _native.unflatten_levels(grad_src_flat.contiguous() if need_src else None,
                                                       grad_pos_flat.contiguous() if (need_pos or need[0]) else None,
                                                       ctx.shapes, want_level_embed=need[0])
        return (g_embed,) + tuple(gs if gs is not None else [None] * L) + tuple(gp if (gp is not None and need_pos) else [None] * L)


def get_valid_ratio(mask):
    """mask[N,H,W] bool (True = padding) -> [N,2] = (valid width / W, valid height / H); the valid extent is
    read off the first column / first row, as the reference does."""
    _, height, width = mask.shape
    valid_h = (~mask[:, :, 0]).sum(1).float() / height
    valid_w = (~mask[:, 0, :]).sum(1).float() / width
    return torch.stack([valid_w, valid_h], -1)


def flatten_feature_levels(srcs, masks, pos_embeds, level_embed):
    """srcs / pos_embeds: lists of [N,C,H_l,W_l]; masks: list of [N,H_l,W_l] bool; level_embed [L,C].
    Returns (src_flatten[N,S,C], mask_flatten[N,S], lvl_pos_embed_flatten[N,S,C], spatial_shapes int64[L,2],
    level_start_index int64[L], valid_ratios[N,L,2])."""
    srcs, masks, pos_embeds = list(srcs), list(masks), list(pos_embeds)
    if (not torch.is_autocast_enabled() and _native.flatten_levels_supported(srcs, pos_embeds, level_embed)):
        # MI355X path: both flattened tensors from one kernel (and one for their gradients)
        shapes = [(s.shape[2], s.shape[3]) for s in srcs]
        src_flatten, pos_flatten = _FlattenLevelsFn.apply(level_embed, *srcs, *pos_embeds)
        spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=src_flatten.device)
        level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
        valid_ratios = torch.stack([get_valid_ratio(m) for m in masks], 1)
        return (src_flatten, torch.cat([m.flatten(1) for m in masks], 1), pos_flatten, spatial_shapes, level_start_index,
                valid_ratios)
    src_parts, mask_parts, pos_parts, shapes = [], [], [], []
    for lvl, (src, mask, pos) in enumerate(zip(srcs, masks, pos_embeds)):
        shapes.append((src.shape[2], src.shape[3]))
        src_parts.append(src.flatten(2).transpose(1, 2))
        mask_parts.append(mask.flatten(1))
        pos_parts.append(pos.flatten(2).transpose(1, 2) + level_embed[lvl].view(1, 1, -1))
    src_flatten = torch.cat(src_parts, 1)
    spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=src_flatten.device)
    level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
    valid_ratios = torch.stack([get_valid_ratio(m) for m in masks], 1)
    return (src_flatten, torch.cat(mask_parts, 1), torch.cat(pos_parts, 1), spatial_shapes, level_start_index,
            valid_ratios)


def encoder_reference_points(spatial_shapes, valid_ratios, device=None):
    """[N, S, L, 2]: for every pixel of every level its centre, normalised by that level's VALID extent, expressed
    in each level's frame (times that level's valid ratio).  ``spatial_shapes`` may be a tensor or a list of
    (H, W); a tensor is read back to the host once (the grid sizes are Python ints, as in the reference)."""
    if torch.is_tensor(spatial_shapes):
        spatial_shapes = [tuple(int(x) for x in hw) for hw in spatial_shapes.tolist()]
    device = valid_ratios.device if device is None else device
    refs = []
    for lvl, (height, width) in enumerate(spatial_shapes):
        ys = torch.linspace(0.5, height - 0.5, height, dtype=torch.float32, device=device)
        xs = torch.linspace(0.5, width - 0.5, width, dtype=torch.float32, device=device)
        ref_y = ys[:, None].expand(height, width).reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * height)
        ref_x = xs[None, :].expand(height, width).reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * width)
        refs.append(torch.stack((ref_x, ref_y), -1))
    return torch.cat(refs, 1)[:, :, None] * valid_ratios[:, None]


def decoder_reference_points(reference_points, valid_ratios):
    """reference_points[N,Lq,2] (or [N,Lq,42]: 21 (x, y) keypoints) -> [N,Lq,L,2] (or [...,42]): the points in
    every level's frame."""
    width = reference_points.shape[-1]
    if width == 42:
        return reference_points[:, :, None] * valid_ratios.repeat(1, 1, 21)[:, None]
    if width != 2:
        raise ValueError("reference_points must have 2 or 42 coordinates per query, got %d" % width)
    return reference_points[:, :, None] * valid_ratios[:, None]
