"""The torch C++ extension over the C ABI (uvhand_amd/_msda_torch.so, csrc/torch_ext/msda_torch.cpp): the C++ autograd node
that MSDeformAttnFunction.apply uses for fp32 / fp64 CUDA tensors computes exactly what the Python class computes through
the ctypes binding (same library, same kernels), keeps the reference's host-side errors, and is what runs by default."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    from uvhand_amd import _ext
    mod = _ext.get()
    assert mod is not None, "uvhand_amd/_msda_torch.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    return mod


def _inputs(case, dtype=torch.float32):
    z = load_golden(case)
    t = lambda k: torch.from_numpy(np.ascontiguousarray(z[k])).cuda()
    f = lambda k: t(k).to(dtype)
    return z, f("value"), t("shapes"), t("level_start"), f("loc"), f("attn"), f("grad_out")


@pytest.mark.parametrize("case,dtype", [("cfg1", torch.float32), ("oob", torch.float32), ("chunk", torch.float64),
                                        ("testpy_grad_D71", torch.float64)])
def test_cpp_node_equals_python_node(ext, case, dtype):
    from torch.autograd import Function
    from uvhand_amd.functions import MSDeformAttnFunction
    z, value, shapes, lsi, loc, attn, go = _inputs(case, dtype)
    res = []
    for use_ext in (True, False):
        v, l, a = (x.clone().requires_grad_(True) for x in (value, loc, attn))
        if use_ext:
            out = MSDeformAttnFunction.apply(v, shapes, lsi, l, a, 2)
            assert "MSDAFunction" in out.grad_fn.name()                     # the C++ node really ran
        else:
            out = Function.apply.__func__(MSDeformAttnFunction, v, shapes, lsi, l, a, 2)     # the Python class itself
            assert "MSDeformAttnFunction" in out.grad_fn.name()
        out.backward(go)
        res.append((out.detach(), v.grad, l.grad, a.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
    assert rel_err(res[0][1].cpu().numpy(), res[1][1].cpu().numpy()) < 2e-5   # grad_value: summation order only
    tol = 1e-11 if dtype == torch.float64 else 5e-6
    assert rel_err(res[0][0].cpu().numpy(), z["out"]) < tol


def test_cpp_entry_points_and_errors(ext):
    z, value, shapes, lsi, loc, attn, go = _inputs("oob")
    out = ext.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 64)
    gv, gl, ga = ext.ms_deform_attn_backward(value, shapes, lsi, loc, attn, go, 64, False)
    assert rel_err(out.cpu().numpy(), z["out"]) < 5e-6 and rel_err(gv.cpu().numpy(), z["grad_value"]) < 2e-5
    with pytest.raises(RuntimeError, match="value tensor has to be contiguous"):
        ext.ms_deform_attn_forward(value.transpose(1, 2), shapes, lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ext.ms_deform_attn_forward(value.cpu(), shapes.cpu(), lsi.cpu(), loc.cpu(), attn.cpu(), 64)
    with pytest.raises(RuntimeError, match=r"batch\(2\) must divide"):
        ext.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 0)
    with pytest.raises(RuntimeError, match="scalar type Long"):
        ext.ms_deform_attn_forward(value, shapes.int(), lsi, loc, attn, 64)
    # deterministic flag reaches the kernels (D = 32 family): bitwise equal twice
    z, value, shapes, lsi, loc, attn, go = _inputs("cfg1")
    a = ext.ms_deform_attn_backward(value, shapes, lsi, loc, attn, go, 64, True)
    b = ext.ms_deform_attn_backward(value, shapes, lsi, loc, attn, go, 64, True)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert rel_err(a[0].cpu().numpy(), z["grad_value"]) < 2e-5


def test_mixed_dtypes_fall_back_to_the_python_node(ext):
    """fp16 value with fp32 locations (the reference's value.to(float32) case) is not the extension's business."""
    from uvhand_amd.functions import MSDeformAttnFunction
    z, value, shapes, lsi, loc, attn, go = _inputs("oob")
    v = value.half().requires_grad_(True)
    out = MSDeformAttnFunction.apply(v, shapes, lsi, loc, attn, 64)
    assert "MSDeformAttnFunction" in out.grad_fn.name() and out.dtype == torch.float32


def test_all_half_inputs_are_computed_in_float32(ext):
    """The amp branch of the dino copy of the module (models/dino/ops/modules/ms_deform_attn.py:124-131) up-casts half
    tensors before the call; here the Function does it itself: float32 inside, gradients back in the inputs' dtypes."""
    from uvhand_amd.functions import MSDeformAttnFunction
    z, value, shapes, lsi, loc, attn, go = _inputs("cfg1")
    v, l, a = (t.half().requires_grad_(True) for t in (value, loc, attn))
    out = MSDeformAttnFunction.apply(v, shapes, lsi, l, a, 64)
    assert out.dtype == torch.float32
    out.backward(go)
    assert v.grad.dtype == l.grad.dtype == a.grad.dtype == torch.float16
    ref = MSDeformAttnFunction.apply(v.detach().float(), shapes, lsi, l.detach().float(), a.detach().float(), 64)
    assert torch.equal(out.detach(), ref)


def test_bf16_cpp_node_equals_python_node(ext):
    from torch.autograd import Function
    from uvhand_amd.functions import MSDeformAttnBF16Function
    z, value, shapes, lsi, loc, attn, go = _inputs("cfg1")
    res = []
    for vdtype in (torch.float32, torch.bfloat16):
        for use_ext in (True, False):
            v = value.to(vdtype).clone().requires_grad_(True)
            l, a = loc.clone().requires_grad_(True), attn.clone().requires_grad_(True)
            if use_ext:
                out = MSDeformAttnBF16Function.apply(v, shapes, lsi, l, a, 64)
                assert "MSDABF16Function" in out.grad_fn.name()
            else:
                out = Function.apply.__func__(MSDeformAttnBF16Function, v, shapes, lsi, l, a, 64)
                assert "MSDeformAttnBF16Function" in out.grad_fn.name()
            assert out.dtype == torch.bfloat16
            out.backward(go.to(torch.bfloat16))
            assert v.grad.dtype == vdtype
            res.append((out.detach().float(), v.grad.float(), l.grad, a.grad))
        x, y = res[-2], res[-1]
        assert torch.equal(x[0], y[0]) and torch.equal(x[2], y[2]) and torch.equal(x[3], y[3])
        assert rel_err(x[1].cpu().numpy(), y[1].cpu().numpy()) < 8e-3          # grad_value: summation order (+ bf16 rounding)
    assert rel_err(res[0][0].cpu().numpy(), z["out"]) < 1e-2                  # bf16 rows vs the fp64 golden
