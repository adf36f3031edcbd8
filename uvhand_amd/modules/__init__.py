from .deformable_layers import (DeformableTransformerDecoder, DeformableTransformerDecoderLayer,
                                DeformableTransformerEncoder, DeformableTransformerEncoderLayer)
from .ms_deform_attn import MSDeformAttn

__all__ = ["MSDeformAttn", "DeformableTransformerEncoderLayer", "DeformableTransformerDecoderLayer",
           "DeformableTransformerEncoder", "DeformableTransformerDecoder"]
