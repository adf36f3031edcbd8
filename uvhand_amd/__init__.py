"""uvhand_amd — MI355X-native multi-scale deformable attention for UVHand.

Drop-in for the reference's ``models/ops`` package surface:

    from uvhand_amd.modules import MSDeformAttn              # models/ops/modules/ms_deform_attn.py
    from uvhand_amd.functions import MSDeformAttnFunction    # models/ops/functions/ms_deform_attn_func.py

Python host code on PyTorch-ROCm, hand-written HIP kernels (``csrc/``) behind the C ABI
declared in ``include/msda.h``.  There is no CPU or pure-PyTorch implementation in this
package: without the built HIP library every call raises.
"""
from ._native import set_exact_nonfinite
from .functions import MSDeformAttnFunction
from .graphs import graphed
from .modules import MSDeformAttn

__all__ = ["MSDeformAttn", "MSDeformAttnFunction", "graphed", "set_exact_nonfinite"]
__version__ = "0.1.0"
