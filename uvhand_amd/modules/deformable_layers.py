"""``DeformableTransformerEncoderLayer`` / ``DeformableTransformerDecoderLayer`` — the transformer layers either side of
the op, drop-in for UVHand ``models/arctic_transformer.py:261-300`` and ``:334-391`` (same classes,
same lines ±6, in ``origin_deformable_transformer.py`` and ``assembly_transformer.py``).

Contract kept: constructor signatures and defaults; sub-module names and creation order (``self_attn, dropout1, norm1,
linear1, dropout2, linear2, dropout3, norm2`` / ``cross_attn, dropout1, norm1, self_attn, dropout2, norm2, linear1,
dropout3, linear2, dropout4, norm3, inter_rp, attn_matrix``), hence state_dict keys and a seeded construction that
consumes the RNG stream like the reference's; ``with_pos_embed`` / ``forward_ffn`` / ``forward`` signatures; the
decoder's ``nn.MultiheadAttention`` fed sequence-first exactly as in :374-376.

What is MI355X-specific: the attention is this package's ``MSDeformAttn`` (HIP kernels); every
``x = x + dropout(x2); x = norm(x)`` pair is one fused add+LayerNorm kernel per direction
(``functions/layernorm_func.py``; the dropout itself stays ``nn.Dropout``, so training uses PyTorch's random stream);
the FFN's two ``nn.Linear`` layers take the split-M MFMA weight-gradient kernel (``functions/linear_func.py``) — with
N*S = 33 440 rows per rank at the training shape the weight gradient is the GEMM the vendor BLAS runs worst.
"""
import torch.nn.functional as F
from torch import nn

from ..functions.layernorm_func import add_layer_norm
from ..functions.linear_func import bracket_linear
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
        # self attention
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        # ffn
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
        src2 = bracket_linear(self.dropout2(self.activation(bracket_linear(src, self.linear1))), self.linear2)
        return add_layer_norm(src, self.dropout3(src2), self.norm2)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        src2 = self.self_attn(self.with_pos_embed(src, pos), reference_points, src, spatial_shapes, level_start_index,
                              padding_mask)
        src = add_layer_norm(src, self.dropout1(src2), self.norm1)
        return self.forward_ffn(src)


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        # cross attention
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        # self attention
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        # ffn
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        # parameter-free taps the reference keeps for forward hooks (:358-359)
        self.inter_rp = nn.ReLU()
        self.attn_matrix = nn.ReLU()

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        tgt2 = bracket_linear(self.dropout3(self.activation(bracket_linear(tgt, self.linear1))), self.linear2)
        return add_layer_norm(tgt, self.dropout4(tgt2), self.norm3)

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index, src_padding_mask=None):
        self.inter_rp(reference_points)
        # self attention over the queries (sequence-first, as the reference feeds nn.MultiheadAttention)
        q = k = self.with_pos_embed(tgt, query_pos)
        tgt2, attn_matrix = self.self_attn(q.transpose(0, 1), k.transpose(0, 1), tgt.transpose(0, 1))
        self.attn_matrix(attn_matrix)
        tgt = add_layer_norm(tgt, self.dropout2(tgt2.transpose(0, 1)), self.norm2)
        # cross attention into the feature pyramid
        tgt2 = self.cross_attn(self.with_pos_embed(tgt, query_pos), reference_points, src, src_spatial_shapes,
                               level_start_index, src_padding_mask)
        tgt = add_layer_norm(tgt, self.dropout1(tgt2), self.norm1)
        return self.forward_ffn(tgt)
