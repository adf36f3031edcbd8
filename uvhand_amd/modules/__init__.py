from .deformable_layers import DeformableTransformerDecoderLayer, DeformableTransformerEncoderLayer
from .ms_deform_attn import MSDeformAttn

__all__ = ["MSDeformAttn", "DeformableTransformerEncoderLayer", "DeformableTransformerDecoderLayer"]
