"""INTEGRATION.md option 2: the reference's own Python driving this package's native module.  The shim directory goes on
sys.path where the CUDA build product `MultiScaleDeformableAttention*.so` would be; a reference-style autograd Function
(written here against the pybind surface of models/ops/src/vision.cpp:13-16, the way
models/ops/functions/ms_deform_attn_func.py:21-39 uses it: value.to(float32), saved un-cast inputs, three gradients
unpacked from the returned list) is then checked against the golden vectors."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from conftest import ROOT, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def MSDA():
    shim_dir = os.path.join(ROOT, "uvhand_amd", "shim")
    sys.path.insert(0, shim_dir)
    try:
        mod = importlib.import_module("MultiScaleDeformableAttention")
    finally:
        sys.path.remove(shim_dir)
    return mod


def _reference_style_function(MSDA):
    class Fn(Function):
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, attn, im2col_step):
            ctx.im2col_step = im2col_step
            out = MSDA.ms_deform_attn_forward(value.to(torch.float32), shapes, lsi, loc, attn, ctx.im2col_step)
            ctx.save_for_backward(value, shapes, lsi, loc, attn)
            return out

        @staticmethod
        @once_differentiable
        def backward(ctx, grad_output):
            value, shapes, lsi, loc, attn = ctx.saved_tensors
            grad_value, grad_loc, grad_attn = MSDA.ms_deform_attn_backward(value.to(torch.float32), shapes, lsi, loc, attn,
                                                                           grad_output, ctx.im2col_step)
            return grad_value, None, None, grad_loc, grad_attn, None
    return Fn


def test_shim_exports_the_pybind_surface(MSDA):
    assert callable(MSDA.ms_deform_attn_forward) and callable(MSDA.ms_deform_attn_backward)
    assert os.path.basename(MSDA.__file__) == "MultiScaleDeformableAttention.py"


@pytest.mark.parametrize("case", ["cfg1", "cfg2_sub", "oob", "chunk", "testpy_float"])
def test_reference_style_function_on_the_shim(MSDA, case):
    z = load_golden(case)
    Fn = _reference_style_function(MSDA)
    t = lambda k: torch.from_numpy(np.ascontiguousarray(z[k])).cuda()
    v, l, a = t("value").requires_grad_(True), t("loc").requires_grad_(True), t("attn").requires_grad_(True)
    out = Fn.apply(v, t("shapes"), t("level_start"), l, a, 2)
    out.backward(t("grad_out"))
    torch.cuda.synchronize()
    assert rel_err(out.detach().cpu().numpy(), z["out"]) < 5e-6
    assert rel_err(v.grad.cpu().numpy(), z["grad_value"]) < 2e-5
    assert rel_err(a.grad.cpu().numpy(), z["grad_attn"]) < 2e-5


def test_shim_backward_returns_a_list_and_keeps_reference_errors(MSDA):
    z = load_golden("oob")
    t = lambda k: torch.from_numpy(np.ascontiguousarray(z[k])).cuda()
    res = MSDA.ms_deform_attn_backward(t("value"), t("shapes"), t("level_start"), t("loc"), t("attn"), t("grad_out"), 64)
    assert isinstance(res, list) and len(res) == 3
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(t("value").cpu(), t("shapes").cpu(), t("level_start").cpu(), t("loc").cpu(), t("attn").cpu(), 64)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDA.ms_deform_attn_forward(t("value").transpose(1, 2), t("shapes"), t("level_start"), t("loc"), t("attn"), 64)
