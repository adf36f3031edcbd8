from .ms_deform_attn_func import (MSDeformAttnBF16Function, MSDeformAttnFunction, MSDeformAttnMergedPrologueFunction,
                                  MSDeformAttnPrologueFunction)

__all__ = ["MSDeformAttnFunction", "MSDeformAttnBF16Function", "MSDeformAttnPrologueFunction",
           "MSDeformAttnMergedPrologueFunction"]
