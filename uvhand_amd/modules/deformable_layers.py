"""``DeformableTransformerEncoderLayer`` / ``DeformableTransformerDecoderLayer`` — the transformer layers either side of
the op, drop-in for UVHand ``models/arctic_transformer.py:261-300`` and ``:334-391`` (same classes,
same lines ±6, in ``origin_deformable_transformer.py`` and ``assembly_transformer.py``) — and the two stacks that call
them, ``DeformableTransformerEncoder`` / ``DeformableTransformerDecoder`` (``:302-330``, ``:394-460``): the layer loops with
the reference-point hand-off between layers.

Contract kept: constructor signatures and defaults; sub-module names and creation order (``self_attn, dropout1, norm1,
linear1, dropout2, linear2, dropout3, norm2`` / ``cross_attn, dropout1, norm1, self_attn, dropout2, norm2, linear1,
dropout3, linear2, dropout4, norm3, inter_rp, attn_matrix``), hence state_dict keys and a seeded construction that
consumes the RNG stream like the reference's; ``with_pos_embed`` / ``forward_ffn`` / ``forward`` signatures; the
decoder's ``nn.MultiheadAttention`` fed sequence-first exactly as in :374-376.

What is MI355X-specific: the attention is this package's ``MSDeformAttn`` (HIP kernels); every
``x = x + dropout(x2); x = norm(x)`` pair is one fused add+LayerNorm kernel per direction
(``functions/layernorm_func.py``; the dropout itself stays ``nn.Dropout``, so training uses PyTorch's random stream);
the FFN ``linear2(dropout(relu(linear1(x))))`` is ONE autograd node (``functions/linear_func.py``: ``fused_ffn``): bias + ReLU
in the epilogue of linear1's GEMM, PyTorch's own dropout kernel (same random stream), the dropout and ReLU gradients as one
in-place pass, and both weight gradients on the split-M MFMA kernel — with N*S = 33 440 rows per rank at the training shape
the weight gradient is the GEMM the vendor BLAS runs worst; the decoder's self-attention keeps the ``nn.MultiheadAttention``
module and its parameters, but unless a hook on ``attn_matrix`` asks for the head-averaged attention matrix (or
``always_attention_matrix``) its core runs on the library's fp32-MFMA attention kernels (``functions/attention_func.py``).
"""
import copy
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..functions.attention_func import self_attention
from ..functions.layernorm_func import add_layer_norm
from ..functions.linear_func import bracket_linear, fused_ffn
from ..utils.transformer_inputs import decoder_reference_points, encoder_reference_points
from .ms_deform_attn import MSDeformAttn


_ACTIVATIONS = {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}


def _get_activation_fn(activation):
    """Name -> functional of the FFN's non-linearity; the same three names and the same exception type as the
    reference's helper (models/arctic_transformer.py:463-471) accept."""
    try:
        return _ACTIVATIONS[activation]
    except KeyError:
        raise RuntimeError("unknown activation %r (one of: %s)" % (activation, ", ".join(sorted(_ACTIVATIONS)))) from None


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, src):
        src2 = fused_ffn(src, self.linear1, self.activation, self.dropout2, self.linear2)
        return add_layer_norm(src, self.dropout3(src2), self.norm2)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        src2 = self.self_attn(self.with_pos_embed(src, pos), reference_points, src, spatial_shapes, level_start_index,
                              padding_mask)
        src = add_layer_norm(src, self.dropout1(src2), self.norm1)
        return self.forward_ffn(src)


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        # two parameter-free modules the reference's forward calls on the reference points / attention matrix (hook points)
        self.inter_rp = nn.ReLU()
        self.attn_matrix = nn.ReLU()
        # True: compute the head-averaged attention matrix on every call, as the reference's forward does, hook or no hook
        # (and with it nn.functional.dropout's random stream for the attention dropout); UVHAND_ALWAYS_ATTENTION_MATRIX=1
        # sets it for every layer built afterwards.  Default: only when a hook on `attn_matrix` listens — 12-layer training
        # step 26.7 -> 25.2 ms (tools/ddp_step.py, profiles/r05_notes.md section 9).
        self.always_attention_matrix = os.environ.get("UVHAND_ALWAYS_ATTENTION_MATRIX", "0") != "0"

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        tgt2 = fused_ffn(tgt, self.linear1, self.activation, self.dropout3, self.linear2)
        return add_layer_norm(tgt, self.dropout4(tgt2), self.norm3)

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index, src_padding_mask=None):
        self.inter_rp(reference_points)
        # the queries attend to each other first (sequence-first, as the reference feeds nn.MultiheadAttention)
        q = k = self.with_pos_embed(tgt, query_pos)
        # The head-averaged attention matrix exists for whoever hooks `attn_matrix`; with no hook registered nothing reads it,
        # and the attention core runs without ever writing a [N*heads, L, L] tensor (same values in eval; in training the
        # attention dropout then draws from that kernel's random stream, not nn.functional.dropout's).
        listened = self.always_attention_matrix or _has_listener(self.attn_matrix)
        if listened:
            tgt2, attn_matrix = self.self_attn(q.transpose(0, 1), k.transpose(0, 1), tgt.transpose(0, 1), need_weights=True)
            self.attn_matrix(attn_matrix)
        else:                                            # the library's attention core where it applies (functions/attention_func.py)
            tgt2 = self_attention(self.self_attn, q.transpose(0, 1), tgt.transpose(0, 1))
        tgt = add_layer_norm(tgt, self.dropout2(tgt2.transpose(0, 1)), self.norm2)
        # then sample the feature pyramid
        tgt2 = self.cross_attn(self.with_pos_embed(tgt, query_pos), reference_points, src, src_spatial_shapes,
                               level_start_index, src_padding_mask)
        tgt = add_layer_norm(tgt, self.dropout1(tgt2), self.norm1)
        return self.forward_ffn(tgt)


def _has_listener(module):
    """Does anything observe calls of this (parameter-free tap) module: its own forward / backward hooks, or hooks registered
    for every module (torch.nn.modules.module.register_module_forward_hook and friends)?"""
    from torch.nn.modules import module as _m
    own = (module._forward_hooks, module._forward_pre_hooks, module._backward_hooks, getattr(module, "_backward_pre_hooks", None))
    glob = (getattr(_m, "_global_forward_hooks", None), getattr(_m, "_global_forward_pre_hooks", None),
            getattr(_m, "_global_backward_hooks", None), getattr(_m, "_global_backward_pre_hooks", None),
            getattr(_m, "_global_forward_hooks_always_called", None))
    return any(bool(h) for h in own + glob)


def _get_clones(module, n):
    """n independent deep copies (models/arctic_transformer.py:459-460): every layer of a stack starts from the same values."""
    return nn.ModuleList(copy.deepcopy(module) for _ in range(n))


def inverse_sigmoid(x, eps=1e-5):
    """logit of x clamped to [0, 1], with both x and 1 - x floored at eps (util/misc.py:614-618)."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class DeformableTransformerEncoder(nn.Module):
    """``num_layers`` copies of an encoder layer over the flattened pyramid (models/arctic_transformer.py:302-330).  The
    reference points — every pixel centre of every level in every level's frame, ``:310-323`` — are built once and shared
    by all layers."""

    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        return encoder_reference_points(spatial_shapes, valid_ratios, device)

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None):
        reference_points = self.get_reference_points(spatial_shapes, valid_ratios, device=src.device)
        output = src
        for layer in self.layers:
            output = layer(output, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return output


class DeformableTransformerDecoder(nn.Module):
    """``num_layers`` copies of a decoder layer (models/arctic_transformer.py:394-457).  Per layer the query reference
    points (2 numbers, or the 21 ARCTIC keypoints = 42) are scaled into every level's valid frame (``:413-419``); when the
    model has attached its per-layer heads (``cls_embed`` / ``key_embed`` / ``obj_key_embed``, set from outside as in the
    reference) the points are refined after each layer and handed on detached (``:423-447``): queries classified as an
    object move by ``obj_key_embed``, queries classified as a hand (classes ``hand_classes``) by ``key_embed``, everything
    else (class 0) stays."""

    hand_classes = (12, 13)                       # left / right hand in the ARCTIC label set (:435)

    def __init__(self, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.cls_embed = None
        self.key_embed = None
        self.obj_key_embed = None

    def _refine(self, lid, output, reference_points):
        classes = self.cls_embed[lid](output).argmax(dim=-1)
        is_hand = torch.zeros_like(classes, dtype=torch.bool)
        for c in self.hand_classes:
            is_hand |= classes == c
        is_obj = ~is_hand & (classes != 0)
        delta_hand, delta_obj = self.key_embed[lid](output), self.obj_key_embed[lid](output)
        unsig = inverse_sigmoid(reference_points)
        unsig = torch.where(is_obj[..., None], unsig + delta_obj, unsig)
        unsig = torch.where(is_hand[..., None], unsig + delta_hand, unsig)
        return (unsig.sigmoid() * 2 - 1).detach()

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None):
        output = tgt
        intermediate, intermediate_reference_points = [], []
        for lid, layer in enumerate(self.layers):
            if reference_points.shape[-1] not in (2, 42):
                raise AssertionError("reference_points must have 2 or 42 coordinates per query")
            reference_points_input = decoder_reference_points(reference_points, src_valid_ratios)
            output = layer(output, query_pos, reference_points_input, src, src_spatial_shapes, src_level_start_index,
                           src_padding_mask)
            if self.cls_embed is not None:
                reference_points = self._refine(lid, output, reference_points)
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points
