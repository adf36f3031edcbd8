"""Pins the oracle (oracle/msda_oracle.c, oracle/torch_fallback.py) to the golden vectors that
tests/golden/gen_golden.py produced by running the reference's own fallback
(UVHand models/ops/functions/ms_deform_attn_func.py:42-62) and autograd through it.  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import OP_CASES, load_golden, near_boundary_mask, rel_err


@pytest.mark.parametrize("case", OP_CASES)
def test_c_oracle_fp64_matches_reference_golden(oracle, case):
    z = load_golden(case)
    args = [z["value"].astype(np.float64), z["shapes"], z["level_start"], z["loc"].astype(np.float64),
            z["attn"].astype(np.float64)]
    out = oracle.forward(*args)
    gv, gl, ga = oracle.backward(z["grad_out"].astype(np.float64), *args)
    assert rel_err(out, z["out"]) < 1e-13
    assert rel_err(gv, z["grad_value"]) < 1e-13
    assert rel_err(ga, z["grad_attn"]) < 1e-12
    if "exact_m1" in z:
        # pixel coordinate == -1 exactly: the reference CUDA kernel drops the point
        # (ms_deform_im2col_cuda.cuh:288) while its grid_sample fallback keeps a zero-weight tap
        # with a non-zero location gradient.  The oracle follows the kernel.
        keep = ~z["exact_m1"]
        assert rel_err(gl[keep], z["grad_loc"][keep]) < 1e-12
        assert np.all(gl[z["exact_m1"]] == 0)
    else:
        assert rel_err(gl, z["grad_loc"]) < 1e-12


@pytest.mark.parametrize("case", OP_CASES)
def test_c_oracle_fp32_close_to_reference_golden(oracle, case):
    z = load_golden(case)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    out = oracle.forward(*args)
    gv, gl, ga = oracle.backward(z["grad_out"], *args)
    assert out.dtype == np.float32
    assert rel_err(out, z["out"]) < 2e-6
    assert rel_err(gv, z["grad_value"]) < 5e-6
    assert rel_err(ga, z["grad_attn"]) < 5e-6
    keep = ~near_boundary_mask(z)
    assert rel_err(gl[keep], z["grad_loc"][keep]) < 5e-6


@pytest.mark.parametrize("case", ["testpy_float", "oob", "edges", "chunk"])
def test_reference_fp32_results_within_its_own_test_tolerance(oracle, case):
    """The reference's float check is allclose(rtol=1e-2, atol=1e-3) (models/ops/test.py:56)."""
    z = load_golden(case)
    out = oracle.forward(z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"])
    assert np.allclose(out, z["out_f32"], rtol=1e-2, atol=1e-3)
    assert rel_err(out, z["out_f32"]) < 2e-6


@pytest.mark.parametrize("case", ["testpy_double", "cfg1", "oob", "edges", "chunk"])
def test_torch_port_matches_reference_golden(case):
    from oracle.torch_fallback import fwd_bwd
    z = load_golden(case)
    t = {k: torch.from_numpy(z[k].astype(np.float64)) for k in ("value", "loc", "attn", "grad_out")}
    out, gv, gl, ga = fwd_bwd(t["value"], torch.from_numpy(z["shapes"]), t["loc"], t["attn"], t["grad_out"])
    assert rel_err(out.numpy(), z["out"]) < 1e-13
    assert rel_err(gv.numpy(), z["grad_value"]) < 1e-13
    assert rel_err(gl.numpy(), z["grad_loc"]) < 1e-12
    assert rel_err(ga.numpy(), z["grad_attn"]) < 1e-12


def _random_case(seed, N=2, M=3, D=5, Lq=7, P=3, shapes=((5, 4), (2, 3)), lo=-0.3, hi=1.3):
    rng = np.random.default_rng(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    lsi = np.concatenate(([0], np.cumsum(shapes.prod(1))[:-1])).astype(np.int64)
    S, L = int(shapes.prod(1).sum()), len(shapes)
    value = rng.random((N, S, M, D))
    loc = rng.random((N, Lq, M, L, P, 2)) * (hi - lo) + lo
    attn = rng.random((N, Lq, M, L, P))
    go = rng.random((N, Lq, M * D))
    return value, shapes, lsi, loc, attn, go


def test_oracle_backward_is_the_gradient_of_its_forward(oracle):
    """Central differences in fp64 (the reference's own backward check is gradcheck, test.py:76)."""
    value, shapes, lsi, loc, attn, go = _random_case(5)
    gv, gl, ga = oracle.backward(go, value, shapes, lsi, loc, attn)
    f = lambda v, l, a: float((oracle.forward(v, shapes, lsi, l, a) * go).sum())
    rng = np.random.default_rng(0)
    eps = 1e-6
    for arr, grad, which in ((value, gv, 0), (loc, gl, 1), (attn, ga, 2)):
        for _ in range(12):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            hi, lo_ = arr.copy(), arr.copy()
            hi[idx] += eps
            lo_[idx] -= eps
            args_hi = [value, loc, attn]
            args_lo = [value, loc, attn]
            args_hi[which], args_lo[which] = hi, lo_
            num = (f(*args_hi) - f(*args_lo)) / (2 * eps)
            assert abs(num - grad[idx]) < 1e-6 * max(1.0, abs(num)), (which, idx, num, grad[idx])


def test_oracle_properties(oracle):
    value, shapes, lsi, loc, attn, go = _random_case(11)
    out = oracle.forward(value, shapes, lsi, loc, attn)
    gv, gl, ga = oracle.backward(go, value, shapes, lsi, loc, attn)
    # linear in value, adjoint pair (forward, grad_value)
    out2 = oracle.forward(2.5 * value, shapes, lsi, loc, attn)
    assert rel_err(out2, 2.5 * out) < 1e-14
    assert abs((out * go).sum() - (value * gv).sum()) < 1e-10 * abs((out * go).sum())
    # out = sum_p attn_p * sample_p  =>  <out, go> = <attn, grad_attn>
    assert abs((out * go).sum() - (attn * ga).sum()) < 1e-10 * abs((out * go).sum())
    # far outside the maps: exact zeros everywhere
    far = loc + 7.0
    assert not oracle.forward(value, shapes, lsi, far, attn).any()
    gv0, gl0, ga0 = oracle.backward(go, value, shapes, lsi, far, attn)
    assert not gv0.any() and not gl0.any() and not ga0.any()
    # thread count does not change the result (fixed accumulation order)
    oracle.set_num_threads(1)
    gv1, _, _ = oracle.backward(go, value, shapes, lsi, loc, attn)
    oracle.set_num_threads(0)
    assert np.array_equal(gv1, gv)


def test_rebuilt_inputs_of_the_big_module_fixture_are_the_ones_the_reference_saw():
    """tests/golden/big_inputs.py regenerates three 4 MB inputs instead of storing them; the fixture carries their checksums."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import big_inputs
    z = big_inputs.module_enc_big_inputs()
    assert np.array_equal(big_inputs.checksums(z), load_golden("module_enc_big")["input_checksums"])
