from .ms_deform_attn_func import MSDeformAttnFunction

__all__ = ["MSDeformAttnFunction"]
