"""Drop-in for the reference's pybind extension module ``MultiScaleDeformableAttention`` (UVHand
``models/ops/src/vision.cpp:13-16``, built by ``models/ops/setup.py:53``, imported as ``MSDA`` at
``models/ops/functions/ms_deform_attn_func.py:18``) — INTEGRATION.md, option 2: keep the reference's Python
(``functions/`` and ``modules/``) untouched and put THIS directory on ``sys.path`` in place of the CUDA build product.

    ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step) -> Tensor
    ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step) -> [grad_value, grad_sampling_loc, grad_attn_weight]

Both forward to the HIP kernels through the C ABI (``uvhand_amd/_native.py`` -> ``libmsda_hip.so``); the checks and error
messages of ``models/ops/src/cuda/ms_deform_attn_cuda.cu:28-52, 93-117`` and ``models/ops/src/ms_deform_attn.h:38,60`` are
reproduced there.  There is no fallback: without the built library the import of ``uvhand_amd._native`` still works, the
first call raises.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:                      # so that `uvhand_amd` resolves when only this directory was put on the path
    sys.path.insert(0, _ROOT)

from uvhand_amd import _native  # noqa: E402


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    return _native.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    # the pybind function returns std::vector<at::Tensor>; the reference unpacks three values from it
    return list(_native.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                                grad_output.contiguous(), im2col_step))
