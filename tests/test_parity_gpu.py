"""Parity of the HIP path (through the C ABI) with the oracle and with the golden vectors captured
from the reference's fallback.  All tests here need the GPU (`-m gpu`)."""
import numpy as np
import pytest
import torch

from conftest import OP_CASES, load_golden, near_boundary_mask, rel_err

pytestmark = pytest.mark.gpu

PATHS = {"auto": -1, "generic": 0}


@pytest.fixture(scope="module")
def native():
    from uvhand_amd import _native
    _native.load()
    assert torch.cuda.is_available()
    return _native


@pytest.fixture(params=["auto", "generic"])
def path(request, native):
    native.force_path(PATHS[request.param])
    yield request.param
    native.force_path(-1)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None and t.is_floating_point():
        t = t.to(dtype)
    return t.cuda()


def run_hip(z, dtype, im2col_step=64):
    from uvhand_amd.functions import MSDeformAttnFunction
    v = dev(z["value"], dtype).requires_grad_(True)
    l = dev(z["loc"], dtype).requires_grad_(True)
    a = dev(z["attn"], dtype).requires_grad_(True)
    out = MSDeformAttnFunction.apply(v, dev(z["shapes"]), dev(z["level_start"]), l, a, im2col_step)
    out.backward(dev(z["grad_out"], dtype))
    torch.cuda.synchronize()
    return [t.detach().cpu().numpy() for t in (out, v.grad, l.grad, a.grad)]


# ---------------------------------------------------------------------------------------------
# golden vectors (reference fallback, fp64)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", OP_CASES)
def test_fp64_matches_reference_golden(native, case):
    z = load_golden(case)
    out, gv, gl, ga = run_hip(z, torch.float64, im2col_step=2)
    assert out.shape == z["out"].shape and gv.shape == z["grad_value"].shape
    assert rel_err(out, z["out"]) < 1e-12
    assert rel_err(gv, z["grad_value"]) < 1e-12
    assert rel_err(ga, z["grad_attn"]) < 1e-11
    keep = ~z["exact_m1"] if "exact_m1" in z else np.ones(z["attn"].shape, bool)
    assert rel_err(gl[keep], z["grad_loc"][keep]) < 1e-11
    if "exact_m1" in z:
        assert np.all(gl[z["exact_m1"]] == 0)       # the CUDA kernel's behaviour (cuh:288)


@pytest.mark.parametrize("case", OP_CASES)
def test_fp32_matches_reference_golden(native, path, case):
    """fp32 tolerance: 5e-6 of the tensor's max magnitude for the forward, 2e-5 for the gradients
    (float atomics / reduction order); the reference's own float check is rtol=1e-2, atol=1e-3
    (models/ops/test.py:56), asserted as well."""
    z = load_golden(case)
    out, gv, gl, ga = run_hip(z, torch.float32)
    assert np.allclose(out, z["out"], rtol=1e-2, atol=1e-3)
    assert rel_err(out, z["out"]) < 5e-6
    assert rel_err(gv, z["grad_value"]) < 2e-5
    assert rel_err(ga, z["grad_attn"]) < 2e-5
    keep = ~near_boundary_mask(z)
    assert rel_err(gl[keep], z["grad_loc"][keep]) < 2e-5


# ---------------------------------------------------------------------------------------------
# seeded inputs vs the C oracle, both kernel families, ragged sizes
# ---------------------------------------------------------------------------------------------
def make_case(seed, N, shapes, M, D, Lq, P, lo=-0.25, hi=1.25):
    g = torch.Generator().manual_seed(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    L = len(shapes)
    S = int(shapes.prod(1).sum())
    lsi = np.concatenate(([0], np.cumsum(shapes.prod(1))[:-1])).astype(np.int64)
    value = (torch.rand(N, S, M, D, generator=g) * 0.01).numpy()
    loc = (torch.rand(N, Lq, M, L, P, 2, generator=g) * (hi - lo) + lo).numpy()
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = (attn / attn.sum((-1, -2), keepdim=True)).numpy()
    go = torch.rand(N, Lq, M * D, generator=g).numpy()
    return dict(value=value, shapes=shapes, level_start=lsi, loc=loc, attn=attn, grad_out=go)


ORACLE_CASES = {
    # name: (N, shapes, M, D, Lq, P)
    "model_small":   (2, [(12, 12), (6, 6), (3, 3), (2, 2)], 8, 32, 37, 4),     # D=32 path, SPLIT=4
    "ragged_items":  (3, [(7, 5), (3, 2)], 5, 32, 13, 3),                       # items % 8 != 0, LP=6
    "one_point":     (1, [(4, 4)], 8, 32, 9, 1),                                # L*P < 4
    "many_queries":  (2, [(16, 16), (8, 8)], 8, 32, 2100, 4),                   # D=32 path, SPLIT=1
    "ragged_split1": (1, [(9, 9), (5, 4), (2, 2)], 7, 32, 4711, 2),             # SPLIT=1, items % 32 != 0
    "d64":           (2, [(8, 8), (4, 4)], 4, 64, 21, 4),
    "d3_odd":        (2, [(5, 3), (2, 2), (1, 1)], 3, 3, 11, 2),
    "d200":          (1, [(6, 6)], 2, 200, 5, 3),
    "five_levels":   (1, [(10, 10), (5, 5), (3, 3), (2, 2), (1, 1)], 8, 32, 50, 2),
    "wide_map":      (1, [(3, 40), (20, 2)], 8, 32, 64, 4),
    "few_queries":   (1, [(40, 40), (20, 20)], 8, 32, 3, 2),                    # more rows than record slots per workgroup
    "flat_levels":   (2, [(9, 9), (9, 9), (8, 10)], 8, 32, 120, 4),             # equal-size levels: no range skew
    # N*M*L >= 256 with one level of 1920 pixels and Lq*P = 1536: the fused backward launch asks for > 64 KiB of
    # dynamic LDS (tp_cap 1920 rows + 4*1536 records) — hipFuncSetAttribute on the fused kernels (round 1 review)
    "one_big_level": (32, [(40, 48)], 8, 32, 384, 4),
}


def _random_geometries(count, seed):
    """Seeded random D=32 geometries: 1-5 levels of 1x1 ... 24x24 pixels (some pyramids, some not), 1-9 heads,
    1-4 points, 1-700 queries, 1-3 batch elements — the workgroup numbering, range dealing, SPLIT choice and
    single / multi-pass plans all depend on these."""
    rng = np.random.RandomState(seed)
    cases = []
    for _ in range(count):
        L = int(rng.randint(1, 6))
        P = int(rng.randint(1, 5))
        if rng.rand() < 0.5:                                            # a pyramid
            h, w = int(rng.randint(4, 25)), int(rng.randint(4, 25))
            shapes = [(max(1, h >> k), max(1, w >> k)) for k in range(L)]
        else:
            shapes = [(int(rng.randint(1, 25)), int(rng.randint(1, 25))) for _ in range(L)]
        M = int(rng.choice([1, 2, 3, 4, 5, 8, 8, 8, 9]))
        Lq = int(rng.choice([1, 2, 7, 33, 100, 300, 301, 450, 700]))
        N = int(rng.randint(1, 4))
        cases.append((N, shapes, M, 32, Lq, P))
    return cases


@pytest.mark.parametrize("idx,case", list(enumerate(_random_geometries(36, 2024))))
def test_random_geometries_match_c_oracle(native, oracle, idx, case):
    z = make_case(100 + idx, *case)
    out, gv, gl, ga = run_hip(z, torch.float32)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(out, oracle.forward(*args)) < 5e-6, case
    assert rel_err(gv, r_gv) < 2e-5, case
    assert rel_err(ga, r_ga) < 2e-5, case
    keep = ~near_boundary_mask(z, tol=1e-5)
    if keep.any():
        assert rel_err(gl[keep], r_gl[keep]) < 2e-5, case


def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy()


def _misaligned_copy(t):
    """Same values, contiguous, but starting one element into its storage (4-byte aligned for fp32 only)."""
    buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
    view = buf[1:].view(t.shape)
    view.copy_(t)
    return view


def test_contiguous_views_at_odd_offsets(native, oracle):
    """Tensors that are contiguous but not 16-byte aligned (views one element into a buffer): fp32 calls are
    served by the element-wise generic kernels with the same results, and so are bf16 rows (fp32 grad_value)."""
    z = make_case(21, *ORACLE_CASES["model_small"])
    s, i = dev(z["shapes"]), dev(z["level_start"])
    v, l, a, go = (_misaligned_copy(dev(z[k])) for k in ("value", "loc", "attn", "grad_out"))
    assert v.data_ptr() % 16 != 0 and v.is_contiguous()
    out = native.ms_deform_attn_forward(v, s, i, l, a, 64)
    gv, gl, ga = native.ms_deform_attn_backward(v, s, i, l, a, go, 64)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(out.cpu().numpy(), oracle.forward(*args)) < 5e-6
    assert rel_err(gv.cpu().numpy(), r_gv) < 2e-5 and rel_err(ga.cpu().numpy(), r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(gl.cpu().numpy()[keep], r_gl[keep]) < 2e-5
    # bf16 rows at a 2-byte offset: generic kernels too (fp32 grad_value; the bf16 grad_value entry is D = 32-family only)
    zr = dict(z, value=_bf16_round(z["value"]), grad_out=_bf16_round(z["grad_out"]))
    v16 = _misaligned_copy(dev(zr["value"]).to(torch.bfloat16))
    go16 = _misaligned_copy(dev(zr["grad_out"]).to(torch.bfloat16))
    assert v16.data_ptr() % 8 != 0
    args = [zr["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    out16 = native.ms_deform_attn_forward(v16, s, i, dev(z["loc"]), dev(z["attn"]), 64)
    assert rel_err(out16.float().cpu().numpy(), oracle.forward(*args)) < 4e-3
    gv, gl, ga = native.ms_deform_attn_backward(v16, s, i, dev(z["loc"]), dev(z["attn"]), go16, 64, fp32_grad_value=True)
    r_gv, r_gl, r_ga = oracle.backward(zr["grad_out"], *args)
    assert rel_err(gv.cpu().numpy(), r_gv) < 2e-5 and rel_err(ga.cpu().numpy(), r_ga) < 2e-5
    with pytest.raises(RuntimeError, match="msda_backward_bf16_gv32 serves"):
        native.ms_deform_attn_backward(v16, s, i, dev(z["loc"]), dev(z["attn"]), go16, 64)
    assert not native.linear_wgrad_supported(_misaligned_copy(torch.zeros(8, 8, device="cuda")), torch.zeros(8, 8, device="cuda"))


def _random_generic_geometries(count, seed):
    rng = np.random.RandomState(seed)
    cases = []
    for _ in range(count):
        L, P = int(rng.randint(1, 5)), int(rng.randint(1, 6))
        shapes = [(int(rng.randint(1, 14)), int(rng.randint(1, 14))) for _ in range(L)]
        D = int(rng.choice([1, 2, 3, 7, 16, 31, 33, 48, 64, 65, 100]))
        cases.append((int(rng.randint(1, 3)), shapes, int(rng.choice([1, 2, 3, 8])), D, int(rng.choice([1, 9, 40, 130])), P))
    return cases


@pytest.mark.parametrize("idx,case", list(enumerate(_random_generic_geometries(16, 99))))
def test_random_generic_geometries_match_c_oracle(native, oracle, idx, case):
    """The generic family (any D; fp32 and fp64) on seeded random geometries."""
    z = make_case(300 + idx, *case)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    for dtype, tol in ((torch.float32, 2e-5), (torch.float64, 1e-11)):
        zz = {k: (v.astype(np.float64) if dtype == torch.float64 and v.dtype == np.float32 else v) for k, v in z.items()}
        a64 = [zz["value"], zz["shapes"], zz["level_start"], zz["loc"], zz["attn"]]
        out, gv, gl, ga = run_hip(zz, dtype)
        r_gv, r_gl, r_ga = oracle.backward(zz["grad_out"], *a64)
        assert rel_err(out, oracle.forward(*a64)) < tol, (case, dtype)
        assert rel_err(gv, r_gv) < tol and rel_err(ga, r_ga) < tol, (case, dtype)
        keep = ~near_boundary_mask(zz, tol=1e-5 if dtype == torch.float32 else 1e-12)
        if keep.any():
            assert rel_err(gl[keep], r_gl[keep]) < tol, (case, dtype)


@pytest.mark.parametrize("idx,case", list(enumerate(_random_generic_geometries(10, 777))))
def test_bf16_rows_on_generic_geometries(native, oracle, idx, case):
    """bf16 rows outside the D = 32 family (any D): generic kernels, fp32 arithmetic and grad_value.  Tolerances as for
    the D = 32 bf16 tests: forward (rounded once to bf16) 4e-3 of max, the fp32 gradients 2e-5; the autograd Function
    returns grad_value in value's dtype."""
    from uvhand_amd.functions import MSDeformAttnBF16Function
    z = make_case(900 + idx, *case)
    z["value"] = _bf16_round(z["value"])
    z["grad_out"] = _bf16_round(z["grad_out"])
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_out = oracle.forward(*args)
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    v16, go16 = dev(z["value"]).to(torch.bfloat16), dev(z["grad_out"]).to(torch.bfloat16)
    s, i, l, a = dev(z["shapes"]), dev(z["level_start"]), dev(z["loc"]), dev(z["attn"])
    out = native.ms_deform_attn_forward(v16, s, i, l, a, 64)
    assert out.dtype == torch.bfloat16 and rel_err(out.float().cpu().numpy(), r_out) < 4e-3, case
    gv, gl, ga = native.ms_deform_attn_backward(v16, s, i, l, a, go16, 64, fp32_grad_value=True)
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(gv.cpu().numpy(), r_gv) < 2e-5 and rel_err(ga.cpu().numpy(), r_ga) < 2e-5, case
    if keep.any():
        assert rel_err(gl.cpu().numpy()[keep], r_gl[keep]) < 2e-5, case
    vq = v16.clone().requires_grad_(True)
    lq, aq = l.clone().requires_grad_(True), a.clone().requires_grad_(True)
    MSDeformAttnBF16Function.apply(vq, s, i, lq, aq, 64).backward(go16)
    assert vq.grad.dtype == torch.bfloat16 and rel_err(vq.grad.float().cpu().numpy(), r_gv) < 4e-3, case
    assert rel_err(aq.grad.cpu().numpy(), r_ga) < 2e-5, case


def test_bf16_generic_and_d32_families_agree(native, oracle):
    """The same bf16 call through both kernel families (msda_force_path): fp32 gradients within summation-order distance,
    the bf16 output within one rounding."""
    z = make_case(31, *ORACLE_CASES["model_small"])
    v16, go16 = dev(z["value"]).to(torch.bfloat16), dev(z["grad_out"]).to(torch.bfloat16)
    s, i, l, a = dev(z["shapes"]), dev(z["level_start"]), dev(z["loc"]), dev(z["attn"])
    res = []
    try:
        for path in (native.PATH_D32, native.PATH_GENERIC):
            native.force_path(path)
            out = native.ms_deform_attn_forward(v16, s, i, l, a, 64)
            res.append((out.float().cpu().numpy(),) + tuple(
                t.cpu().numpy() for t in native.ms_deform_attn_backward(v16, s, i, l, a, go16, 64, fp32_grad_value=True)))
    finally:
        native.force_path(-1)
    assert rel_err(res[0][0], res[1][0]) < 4e-3
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(res[0][1], res[1][1]) < 2e-5 and rel_err(res[0][3], res[1][3]) < 2e-5
    assert rel_err(res[0][2][keep], res[1][2][keep]) < 2e-5


PILED = {"cfg2_decoder": (2, [(48, 48), (24, 24), (12, 12), (6, 6)], 8, 32, 300, 4),
         # Lq*P beyond one pass: the kept-taps single pass (bwd_value_wide_body) and, when a workgroup's rows receive more
         # taps than its record array holds (small spreads), its chunked fall-back
         "many_queries": (2, [(16, 16), (8, 8)], 8, 32, 2100, 4),
         "long_encoder": (1, [(40, 40), (20, 20), (10, 10), (5, 5)], 8, 32, 2125, 4),
         # Lq*P = 68 000 > 65 536: more than one attempt (16-bit chunk-relative list) and far more taps per workgroup than
         # its record array holds at any spread (9 ranges where 43 would be needed): the count-sized chunks
         "beyond_16bit": (1, [(12, 12)], 2, 32, 17000, 4),
         # P and L*P not powers of two on the kept-taps path (the scan's incremental (q, p) arithmetic, general division)
         "odd_points": (1, [(20, 20), (10, 10), (5, 5)], 4, 32, 700, 3)}


@pytest.mark.parametrize("geometry", list(PILED))
@pytest.mark.parametrize("spread", [0.0, 0.02, 0.2, 0.6, 1.4])
def test_taps_piled_on_few_pixels(native, oracle, spread, geometry):
    """grad_value's counting sort with every sampling point inside a small patch (collisions: thousands of
    taps on one pixel, most rows of the map empty); compared with the C oracle."""
    z = make_case(11, *PILED[geometry])
    g = torch.Generator().manual_seed(12)
    centre = torch.tensor([0.37, 0.61])
    z["loc"] = (centre + (torch.rand(z["loc"].shape, generator=g) - 0.5) * spread).numpy().astype(np.float32)
    out, gv, gl, ga = run_hip(z, torch.float32)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(out, oracle.forward(*args)) < 5e-6
    assert rel_err(gv, r_gv) < 2e-5                     # up to 9 600 taps on one pixel: fp32 summation order
    assert rel_err(ga, r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(gl[keep], r_gl[keep]) < 2e-5


@pytest.mark.parametrize("name", list(ORACLE_CASES))
def test_fp32_matches_c_oracle(native, oracle, path, name):
    z = make_case(1, *ORACLE_CASES[name])
    out, gv, gl, ga = run_hip(z, torch.float32)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_out = oracle.forward(*args)
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(out, r_out) < 5e-6
    assert rel_err(gv, r_gv) < 2e-5
    assert rel_err(ga, r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(gl[keep], r_gl[keep]) < 2e-5


@pytest.mark.parametrize("name", ["model_small", "d3_odd", "d200"])
def test_fp64_matches_c_oracle(native, oracle, name):
    z = make_case(2, *ORACLE_CASES[name])
    z = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in z.items()}
    out, gv, gl, ga = run_hip(z, torch.float64)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(out, oracle.forward(*args)) < 1e-13
    assert rel_err(gv, r_gv) < 1e-12
    assert rel_err(gl, r_gl) < 1e-12
    assert rel_err(ga, r_ga) < 1e-12


# ---------------------------------------------------------------------------------------------
# the reference's own test, restated: models/ops/test.py:31-86
# ---------------------------------------------------------------------------------------------
def _testpy_inputs(channels=2):
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long).cuda()
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = 30
    value = torch.rand(N, S, M, channels).cuda() * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2).cuda()
    attn = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, shapes, lsi, loc, attn


def test_reference_test_forward_double_and_float(native, oracle):
    from uvhand_amd.functions import MSDeformAttnFunction
    torch.manual_seed(3)
    value, shapes, lsi, loc, attn = _testpy_inputs()
    ref = oracle.forward(value.double().cpu().numpy(), shapes.cpu().numpy(), lsi.cpu().numpy(),
                         loc.double().cpu().numpy(), attn.double().cpu().numpy())
    out = MSDeformAttnFunction.apply(value.double(), shapes, lsi, loc.double(), attn.double(), 2)
    assert torch.allclose(out.cpu(), torch.from_numpy(ref))                    # test.py:40 defaults
    golden = load_golden("testpy_double")                                      # same seed, same draws
    assert np.array_equal(golden["value"], value.cpu().numpy())
    assert torch.allclose(out.cpu(), torch.from_numpy(golden["out"]))
    value, shapes, lsi, loc, attn = _testpy_inputs()
    ref = oracle.forward(value.cpu().numpy(), shapes.cpu().numpy(), lsi.cpu().numpy(),
                         loc.cpu().numpy(), attn.cpu().numpy())
    out = MSDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 2)
    assert torch.allclose(out.cpu(), torch.from_numpy(ref), rtol=1e-2, atol=1e-3)   # test.py:56


@pytest.mark.parametrize("channels", [30, 32, 64, 71, 1025, 2048, 3096])
def test_reference_test_gradcheck(native, channels):
    """check_gradient_numerical (test.py:63-78): fp64 gradcheck with torch's defaults."""
    from uvhand_amd.functions import MSDeformAttnFunction
    torch.manual_seed(3 + channels)
    value, shapes, lsi, loc, attn = _testpy_inputs(channels)
    value.requires_grad = True
    loc.requires_grad = True
    attn.requires_grad = True
    assert torch.autograd.gradcheck(
        MSDeformAttnFunction.apply,
        (value.double(), shapes, lsi, loc.double(), attn.double(), 2))


# ---------------------------------------------------------------------------------------------
# edge cases and error behaviour
# ---------------------------------------------------------------------------------------------
def test_empty_and_degenerate_inputs(native, path):
    from uvhand_amd.functions import MSDeformAttnFunction
    shapes = torch.tensor([[4, 4], [2, 2]], dtype=torch.long).cuda()
    lsi = torch.tensor([0, 16], dtype=torch.long).cuda()
    for N, Lq in ((2, 0), (0, 5)):
        v = torch.rand(N, 20, 8, 32).cuda().requires_grad_(True)
        l = torch.rand(N, Lq, 8, 2, 4, 2).cuda().requires_grad_(True)
        a = torch.rand(N, Lq, 8, 2, 4).cuda().requires_grad_(True)
        out = MSDeformAttnFunction.apply(v, shapes, lsi, l, a, 64)
        assert tuple(out.shape) == (N, Lq, 256)
        out.sum().backward()
        torch.cuda.synchronize()
        assert not v.grad.any() and tuple(l.grad.shape) == tuple(l.shape)
    # every location far outside: exact zeros, and NaN locations behave like "outside"
    v = torch.rand(1, 20, 8, 32).cuda().requires_grad_(True)
    l = (torch.rand(1, 6, 8, 2, 4, 2).cuda() + 5.0)
    l[0, 0, 0, 0, 0, 0] = float("nan")
    l.requires_grad_(True)
    a = torch.rand(1, 6, 8, 2, 4).cuda().requires_grad_(True)
    out = MSDeformAttnFunction.apply(v, shapes, lsi, l, a, 64)
    out.backward(torch.ones_like(out))
    torch.cuda.synchronize()
    for t in (out, v.grad, l.grad, a.grad):
        assert not t.isnan().any() and not t.any()


def test_inf_in_value_outside_taps_does_not_leak(native, path):
    """A tap that is not sampled must not be read into the result (0 * inf)."""
    from uvhand_amd.functions import MSDeformAttnFunction
    shapes = torch.tensor([[3, 3]], dtype=torch.long).cuda()
    lsi = torch.tensor([0], dtype=torch.long).cuda()
    v = torch.rand(1, 9, 8, 32).cuda()
    v[0, 0] = float("inf")                      # pixel (0,0)
    l = torch.full((1, 4, 8, 1, 2, 2), 0.75).cuda()   # pixel coordinate 1.75: taps (1,1)..(2,2)
    l[..., 1, :] = 1.2                          # second point partly outside on the far side
    a = torch.full((1, 4, 8, 1, 2), 0.5).cuda()
    out = MSDeformAttnFunction.apply(v, shapes, lsi, l, a, 64)
    assert torch.isfinite(out).all()


def test_host_errors_follow_the_reference(native):
    from uvhand_amd.functions import MSDeformAttnFunction
    z = make_case(0, 4, [(4, 4), (2, 2)], 8, 32, 6, 2)
    v, l, a = dev(z["value"]), dev(z["loc"]), dev(z["attn"])
    s, i = dev(z["shapes"]), dev(z["level_start"])
    with pytest.raises(RuntimeError, match=r"batch\(4\) must divide im2col_step\(3\)"):
        MSDeformAttnFunction.apply(v, s, i, l, a, 3)              # ms_deform_attn_cuda.cu:52
    with pytest.raises(RuntimeError, match="sampling_loc tensor has to be contiguous"):
        MSDeformAttnFunction.apply(v, s, i, l.transpose(1, 2).contiguous().transpose(1, 2), a, 64)
    with pytest.raises(RuntimeError, match="spatial_shapes must be a CUDA tensor"):
        MSDeformAttnFunction.apply(v, s.cpu(), i, l, a, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDeformAttnFunction.apply(v.cpu(), s, i, l, a, 64)
    with pytest.raises(RuntimeError):                              # dtype mismatch (data<scalar_t>())
        MSDeformAttnFunction.apply(v, s, i, l.double(), a, 64)
    # results do not depend on im2col_step; a non-contiguous grad_output is accepted
    v.requires_grad_(True)
    o1 = MSDeformAttnFunction.apply(v, s, i, l, a, 2)
    o2 = MSDeformAttnFunction.apply(v, s, i, l, a, 64)
    assert torch.equal(o1, o2)
    go = torch.rand(256, 6, 4).cuda().permute(2, 1, 0)
    o2.backward(go)
    assert v.grad is not None and torch.isfinite(v.grad).all()


def test_value_is_cast_like_the_reference(native):
    """functions/ms_deform_attn_func.py:26,37: value.to(float32) in forward and backward."""
    from uvhand_amd.functions import MSDeformAttnFunction
    z = make_case(0, 2, [(4, 4), (2, 2)], 8, 32, 6, 2)
    v16 = dev(z["value"]).half().requires_grad_(True)
    l, a = dev(z["loc"]), dev(z["attn"])
    out = MSDeformAttnFunction.apply(v16, dev(z["shapes"]), dev(z["level_start"]), l, a, 64)
    assert out.dtype == torch.float32
    out.sum().backward()
    assert v16.grad.dtype == torch.float16
    ref = MSDeformAttnFunction.apply(v16.detach().float(), dev(z["shapes"]), dev(z["level_start"]), l, a, 64)
    assert torch.equal(out, ref)


# ---------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: size-independent properties (the oracle would take too long)
# ---------------------------------------------------------------------------------------------
FULL = {
    "cfg2_decoder": (2, [(48, 48), (24, 24), (12, 12), (6, 6)], 8, 32, 300, 4),
    "cfg2_encoder": (2, [(48, 48), (24, 24), (12, 12), (6, 6)], 8, 32, 3060, 4),
    "cfg4_decoder": (32, [(28, 28), (14, 14), (7, 7), (4, 4)], 8, 32, 300, 4),
    "cfg4_encoder": (32, [(28, 28), (14, 14), (7, 7), (4, 4)], 8, 32, 1045, 4),
}


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_properties(native, name):
    from uvhand_amd.functions import MSDeformAttnFunction
    z = make_case(0, *FULL[name], lo=-0.1, hi=1.1)
    s, i = dev(z["shapes"]), dev(z["level_start"])
    v = dev(z["value"]).requires_grad_(True)
    l = dev(z["loc"]).requires_grad_(True)
    a = dev(z["attn"]).requires_grad_(True)
    go = dev(z["grad_out"])
    out = MSDeformAttnFunction.apply(v, s, i, l, a, 64)
    out.backward(go)
    torch.cuda.synchronize()
    dot = (out.double() * go.double()).sum().item()
    # the op is linear in value: <out, go> = <value, grad_value>
    assert abs(dot - (v.double() * v.grad.double()).sum().item()) < 1e-5 * abs(dot)
    # out = sum_p attn_p * sample_p: <out, go> = <attn, grad_attn>
    assert abs(dot - (a.double() * a.grad.double()).sum().item()) < 1e-5 * abs(dot)
    # linearity: f(2.5 v) = 2.5 f(v)
    out2 = MSDeformAttnFunction.apply(2.5 * v.detach(), s, i, l.detach(), a.detach(), 64)
    assert rel_err(out2.cpu().numpy(), 2.5 * out.detach().cpu().numpy()) < 1e-6
    # a constant map samples to (sum of in-range bilinear weights) <= 1, and exactly the sum of
    # attention weights for points well inside every level
    ones = torch.ones_like(v.detach())
    inner = l.detach().clamp(0.2, 0.8)
    o = MSDeformAttnFunction.apply(ones, s, i, inner, a.detach(), 64)
    assert rel_err(o.cpu().numpy(), np.ones(o.shape, np.float32)) < 1e-5
    # directional derivative vs central differences, along the direction sign(grad) so that the
    # analytic value is a sum of magnitudes (no cancellation).  The op is piecewise bilinear in the
    # location: a point within eps*W pixels of a pixel centre crosses a kink and contributes an O(1)
    # relative error, i.e. a few percent of the points on the 48-pixel level at eps = 1e-4.
    dl = 0.5 * torch.sign(l.grad)
    eps = 1e-4
    fp = (MSDeformAttnFunction.apply(v.detach(), s, i, l.detach() + eps * dl, a.detach(), 64).double() * go).sum()
    fm = (MSDeformAttnFunction.apply(v.detach(), s, i, l.detach() - eps * dl, a.detach(), 64).double() * go).sum()
    num = ((fp - fm) / (2 * eps)).item()
    ana = (l.grad.double() * dl.double()).sum().item()
    assert ana > 0 and abs(num - ana) < 5e-2 * ana


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_vs_oracle_subsample(native, oracle, name):
    """The bench's own shapes at full size against the C oracle (about a second each on the host): cfg-2 decoder
    (fixed-capacity sort), cfg-2 / cfg-4 encoder (one pass by kept taps, balanced gather), cfg-4 decoder (prefix-sum sort,
    balanced gather)."""
    z = make_case(0, *FULL[name])
    out, gv, gl, ga = run_hip(z, torch.float32)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(out, oracle.forward(*args)) < 5e-6
    assert rel_err(gv, r_gv) < 2e-5
    assert rel_err(ga, r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(gl[keep], r_gl[keep]) < 2e-5


def test_runs_on_the_current_stream_without_sync(native):
    """Stream-ordered: results are right when launched on a side stream."""
    from uvhand_amd.functions import MSDeformAttnFunction
    z = make_case(3, 2, [(8, 8), (4, 4)], 8, 32, 40, 4)
    ref = run_hip(z, torch.float32)[0]
    st = torch.cuda.Stream()
    v, l, a = dev(z["value"]), dev(z["loc"]), dev(z["attn"])
    s, i = dev(z["shapes"]), dev(z["level_start"])
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        out = MSDeformAttnFunction.apply(v, s, i, l, a, 64)
    st.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)


# ---------------------------------------------------------------------------------------------
# bf16 storage (BASELINE config 3): new capability, judged against the fp32/fp64 oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["model_small", "ragged_items", "many_queries", "cfg2_decoder"])
def test_bf16_storage_matches_oracle_on_rounded_inputs(native, oracle, name):
    """Tolerances (stated): with value and grad_out pre-rounded to bf16 the only
    differences from the fp32 oracle are fp32 summation order and the single final rounding to bf16
    (relative 2^-9 = 0.2 %): forward and grad_value within 4e-3 of the tensor's max, grad_loc and
    grad_attn (stored in fp32) within 2e-5."""
    from uvhand_amd.functions import MSDeformAttnBF16Function
    case = ORACLE_CASES[name] if name in ORACLE_CASES else FULL[name]
    z = make_case(4, *case)
    z["value"] = _bf16_round(z["value"])
    z["grad_out"] = _bf16_round(z["grad_out"])
    v = dev(z["value"]).to(torch.bfloat16).requires_grad_(True)
    l = dev(z["loc"]).requires_grad_(True)
    a = dev(z["attn"]).requires_grad_(True)
    out = MSDeformAttnBF16Function.apply(v, dev(z["shapes"]), dev(z["level_start"]), l, a, 64)
    assert out.dtype == torch.bfloat16
    out.backward(dev(z["grad_out"]).to(torch.bfloat16))
    torch.cuda.synchronize()
    assert v.grad.dtype == torch.bfloat16 and l.grad.dtype == torch.float32
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_out = oracle.forward(*args)
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(out.detach().float().cpu().numpy(), r_out) < 4e-3
    assert rel_err(v.grad.float().cpu().numpy(), r_gv) < 4e-3
    assert rel_err(a.grad.cpu().numpy(), r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(l.grad.cpu().numpy()[keep], r_gl[keep]) < 2e-5


@pytest.mark.parametrize("name", ["model_small", "many_queries", "cfg2_decoder"])
def test_bf16_rows_with_fp32_grad_value(native, oracle, name):
    """msda_backward_bf16_gv32: bf16 value / grad_out, float32 grad_value — no rounding of the result, so on
    bf16-representable inputs all three gradients sit within fp32 summation-order distance of the oracle
    (2e-5 of max), single-pass and multi-pass ("many_queries") alike."""
    case = ORACLE_CASES[name] if name in ORACLE_CASES else FULL[name]
    z = make_case(6, *case)
    z["value"] = _bf16_round(z["value"])
    z["grad_out"] = _bf16_round(z["grad_out"])
    gv, gl, ga = native.ms_deform_attn_backward(
        dev(z["value"]).to(torch.bfloat16), dev(z["shapes"]), dev(z["level_start"]), dev(z["loc"]), dev(z["attn"]),
        dev(z["grad_out"]).to(torch.bfloat16), 64, fp32_grad_value=True)
    assert gv.dtype == torch.float32
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(gv.cpu().numpy(), r_gv) < 2e-5
    assert rel_err(ga.cpu().numpy(), r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(gl.cpu().numpy()[keep], r_gl[keep]) < 2e-5
    with pytest.raises(RuntimeError, match="bfloat16 rows only"):
        native.ms_deform_attn_backward(dev(z["value"]), dev(z["shapes"]), dev(z["level_start"]), dev(z["loc"]),
                                       dev(z["attn"]), dev(z["grad_out"]), 64, fp32_grad_value=True)


@pytest.mark.parametrize("idx,case", list(enumerate(_random_geometries(12, 4242))))
def test_bf16_random_geometries(native, oracle, idx, case):
    """bf16 rows on seeded random D=32 geometries (both grad_value variants), same tolerances as above."""
    z = make_case(500 + idx, *case)
    z["value"] = _bf16_round(z["value"])
    z["grad_out"] = _bf16_round(z["grad_out"])
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_out = oracle.forward(*args)
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    v16, go16 = dev(z["value"]).to(torch.bfloat16), dev(z["grad_out"]).to(torch.bfloat16)
    s, i, l, a = dev(z["shapes"]), dev(z["level_start"]), dev(z["loc"]), dev(z["attn"])
    out = native.ms_deform_attn_forward(v16, s, i, l, a, 64)
    assert rel_err(out.float().cpu().numpy(), r_out) < 4e-3, case
    keep = ~near_boundary_mask(z, tol=1e-5)
    for gv32, tol in ((False, 4e-3), (True, 2e-5)):
        gv, gl, ga = native.ms_deform_attn_backward(v16, s, i, l, a, go16, 64, fp32_grad_value=gv32)
        assert rel_err(gv.float().cpu().numpy(), r_gv) < tol, (case, gv32)
        assert rel_err(ga.cpu().numpy(), r_ga) < 2e-5, (case, gv32)
        if keep.any():
            assert rel_err(gl.cpu().numpy()[keep], r_gl[keep]) < 2e-5, (case, gv32)


def test_bf16_storage_vs_fp32_path_and_errors(native):
    from uvhand_amd.functions import MSDeformAttnBF16Function, MSDeformAttnFunction
    z = make_case(5, *ORACLE_CASES["model_small"])
    s, i = dev(z["shapes"]), dev(z["level_start"])
    v, l, a = dev(z["value"]), dev(z["loc"]), dev(z["attn"])
    o32 = MSDeformAttnFunction.apply(v, s, i, l, a, 64)
    o16 = MSDeformAttnBF16Function.apply(v, s, i, l, a, 64)         # fp32 value is rounded on entry
    # bf16 rounding of value (2^-9) and of the result: rtol 2e-2, atol 1e-3*max|out|
    assert torch.allclose(o16.float(), o32, rtol=2e-2, atol=1e-3 * o32.abs().max().item())
    # an fp32 `value` gets its gradient in fp32 straight from the kernel (no bf16 round trip)
    v32 = v.clone().requires_grad_(True)
    MSDeformAttnBF16Function.apply(v32, s, i, l, a, 64).float().sum().backward()
    vref = v.clone().requires_grad_(True)
    MSDeformAttnFunction.apply(vref, s, i, l, a, 64).sum().backward()
    assert v32.grad.dtype == torch.float32
    assert rel_err(v32.grad.cpu().numpy(), vref.grad.cpu().numpy()) < 1e-5      # same weights, grad_out = 1 exactly
    # D != 32: the generic kernels serve bf16 rows (test_bf16_rows_on_generic_geometries); only the bf16-grad_value
    # ABI entry is refused there, loudly
    zz = make_case(5, *ORACLE_CASES["d64"])
    o64 = MSDeformAttnBF16Function.apply(dev(zz["value"]), dev(zz["shapes"]), dev(zz["level_start"]),
                                         dev(zz["loc"]), dev(zz["attn"]), 64)
    r64 = MSDeformAttnFunction.apply(dev(zz["value"]), dev(zz["shapes"]), dev(zz["level_start"]), dev(zz["loc"]), dev(zz["attn"]), 64)
    assert o64.dtype == torch.bfloat16 and torch.allclose(o64.float(), r64, rtol=2e-2, atol=1e-3 * r64.abs().max().item())
    with pytest.raises(RuntimeError, match="msda_backward_bf16_gv32 serves"):
        native.ms_deform_attn_backward(dev(zz["value"]).bfloat16(), dev(zz["shapes"]), dev(zz["level_start"]), dev(zz["loc"]),
                                       dev(zz["attn"]), dev(zz["grad_out"]).bfloat16(), 64)


def test_beyond_int32_element_offsets_uses_the_generic_kernels(native):
    """N*S*M*D = 2^31 elements: the D=32 kernels' 32-bit row offsets no longer cover the tensor, the
    dispatcher must route to the generic family (64-bit offsets) and the far end of the map must work."""
    from uvhand_amd.functions import MSDeformAttnFunction
    H, W, M, D = 2048, 4096, 8, 32                                     # S = 2^23 pixels, 8 GiB of fp32 value
    shapes = torch.tensor([[H, W]], dtype=torch.long).cuda()
    lsi = torch.zeros(1, dtype=torch.long).cuda()
    value = torch.zeros(1, H * W, M, D, device="cuda")
    y, x = H - 2, W - 3                                                  # a pixel near the end of the tensor
    value[0, y * W + x] = torch.arange(M * D, dtype=torch.float32, device="cuda").view(M, D)
    value.requires_grad_(True)
    loc = torch.empty(1, 2, M, 1, 1, 2, device="cuda")
    loc[..., 0] = (x + 0.5) / W                                          # exactly the pixel centre
    loc[..., 1] = (y + 0.5) / H
    attn = torch.ones(1, 2, M, 1, 1, device="cuda")
    out = MSDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 64)
    assert torch.equal(out[0, 0], torch.arange(M * D, dtype=torch.float32, device="cuda"))
    out.sum().backward()
    torch.cuda.synchronize()
    g = value.grad[0, y * W + x]
    assert torch.all(g == 2.0) and value.grad.sum().item() == 2.0 * M * D     # two queries hit that pixel, nothing else
    # the deterministic flag on the same geometry: the destination-major kernel walks 2^26 (pixel, head) rows, more
    # wavefronts than one launch holds threads for — it strides over them
    gv, gl, ga = native.ms_deform_attn_backward(value.detach(), shapes, lsi, loc, attn, torch.ones_like(out.detach()), 64,
                                                deterministic=True)
    torch.cuda.synchronize()
    assert torch.all(gv[0, y * W + x] == 2.0) and gv.sum().item() == 2.0 * M * D
    del value, out, gv
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------
# the forward's point table (msda_forward_ws_* / MSDA_FLAG_FORWARD_TABLE): small problems only
# ---------------------------------------------------------------------------------------------
SMALL_TABLE_CASES = {
    "cfg2_decoder": (2, [(48, 48), (24, 24), (12, 12), (6, 6)], 8, 32, 300, 4),
    "model_small":  ORACLE_CASES["model_small"],
    "ragged_items": ORACLE_CASES["ragged_items"],           # items % 8 != 0, M = 5: the table's q / m split by division
    "one_point":    ORACLE_CASES["one_point"],
    "few_queries":  ORACLE_CASES["few_queries"],            # more rows than record slots per workgroup: the general body
    "flat_levels":  ORACLE_CASES["flat_levels"],
    "Lq_384":       (1, [(20, 20), (10, 10)], 8, 32, 384, 4),   # the pair's grad_out rows just fit the LDS stage (<= 399)
    "Lq_400_p2":    (1, [(20, 20), (10, 10)], 8, 32, 400, 2),   # ... and just do not: rows gathered from global memory
}


@pytest.mark.parametrize("name", list(SMALL_TABLE_CASES))
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_backward_from_the_forward_table_equals_backward_from_a_scan(native, oracle, name, dtype):
    """The forward of a small problem can leave its per-point table for the backward (include/msda.h: msda_forward_ws_*); role B
    then reads 16-byte entries instead of re-deriving them from sampling_loc / attn_weight.  Same records either way:
    grad_sampling_loc / grad_attn_weight bit-identical, grad_value equal up to the order of a row's sum; both against the
    oracle.  Geometries whose backward reads no table report a size of 0 and get None."""
    z = make_case(11, *SMALL_TABLE_CASES[name])
    bf16 = dtype == "bf16"
    if bf16:
        z["value"], z["grad_out"] = _bf16_round(z["value"]), _bf16_round(z["grad_out"])
    rows = torch.bfloat16 if bf16 else torch.float32
    v, go = dev(z["value"]).to(rows), dev(z["grad_out"]).to(rows)
    s, i, l, a = dev(z["shapes"]), dev(z["level_start"]), dev(z["loc"]), dev(z["attn"])
    N, S, M, D = z["value"].shape
    Lq, L, P = z["loc"].shape[1], z["loc"].shape[3], z["loc"].shape[4]
    plan = native.describe_plan(N, S, M, D, L, Lq, P, row_bytes=2 if bf16 else 4)
    out_plain = native.ms_deform_attn_forward(v, s, i, l, a, 64)
    out, table = native.ms_deform_attn_forward(v, s, i, l, a, 64, with_table=True)
    assert torch.equal(out, out_plain)
    assert (table is not None) == ("fixed" in plan), plan
    base = native.ms_deform_attn_backward(v, s, i, l, a, go, 64)
    if table is not None:
        assert table.numel() >= N * M * L * Lq * P * 16                      # (+ the range header behind the point entries)
        got = native.ms_deform_attn_backward(v, s, i, l, a, go, 64, table=table)
        assert torch.equal(got[1], base[1]) and torch.equal(got[2], base[2])
        assert rel_err(got[0].float().cpu().numpy(), base[0].float().cpu().numpy()) < (4e-3 if bf16 else 1e-6)
    else:
        got = base
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"])
    assert rel_err(got[0].float().cpu().numpy(), r_gv) < (4e-3 if bf16 else 2e-5)
    assert rel_err(got[2].cpu().numpy(), r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(got[1].cpu().numpy()[keep], r_gl[keep]) < 2e-5


def test_forward_table_of_a_pile_up_takes_the_general_body(native, oracle):
    """Taps piled on a few pixels overflow the small body's fixed-capacity record slots: the workgroup starts over in the
    general single-pass body — with the table in hand as well as without."""
    z = make_case(12, 2, [(12, 12), (6, 6)], 8, 32, 300, 4)
    z["loc"][..., 0] = 0.31 + 0.02 * z["loc"][..., 0]                     # every point within one or two pixels
    z["loc"][..., 1] = 0.62 + 0.02 * z["loc"][..., 1]
    v, go = dev(z["value"]), dev(z["grad_out"])
    s, i, l, a = dev(z["shapes"]), dev(z["level_start"]), dev(z["loc"]), dev(z["attn"])
    out, table = native.ms_deform_attn_forward(v, s, i, l, a, 64, with_table=True)
    assert table is not None
    got = native.ms_deform_attn_backward(v, s, i, l, a, go, 64, table=table)
    base = native.ms_deform_attn_backward(v, s, i, l, a, go, 64)
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"])
    for g in (got, base):
        assert rel_err(g[0].cpu().numpy(), r_gv) < 2e-5
        assert rel_err(g[2].cpu().numpy(), r_ga) < 2e-5
    assert torch.equal(got[1], base[1])


# ---------------------------------------------------------------------------------------------
# either side of every dispatcher threshold (msda_d32.hip: plan_lds, pick_split, plan_value, plan_fused)
# ---------------------------------------------------------------------------------------------
_S2 = [(8, 8), (4, 4)]
_S5 = [(8, 8), (4, 4), (2, 2), (2, 2), (1, 1)]                # L*P = 20 > 16: never the LDS stage, so pick_split decides
_C2 = [(48, 48), (24, 24), (12, 12), (6, 6)]
STRADDLE = {
    # name: ((N, shapes, M, D, Lq, P), substrings the launch plan must contain)
    "items_32760":   ((1, _S2, 8, 32, 4095, 4), ["fwd=tiled(", "bwd=fused("]),           # below plan_lds' 32 768 items
    "items_32776":   ((1, _S2, 8, 32, 4097, 4), ["fwd=lds(", "bwd=fused_lds("]),         # above
    "octets_2048":   ((1, _S2, 8, 32, 2048, 4), ["fwd=tiled(split=4"]),                  # pick_split: 4 wavefronts per octet ...
    "octets_2049":   ((1, _S2, 8, 32, 2049, 4), ["fwd=tiled(split=2"]),                  # ... 2
    "octets_8192":   ((1, _S5, 8, 32, 8192, 4), ["fwd=tiled(split=2"]),
    "octets_8193":   ((1, _S5, 8, 32, 8193, 4), ["fwd=tiled(split=1"]),                  # ... 1
    "points_1536":   ((1, _S2, 8, 32, 384, 4), ["acc=single", "fixed"]),                 # Lq*P = kSingleMaxPoints: one pass, short sort
    "points_1540":   ((1, _S2, 8, 32, 385, 4), ["acc=wide", "prefix"]),                  # beyond: the kept-taps pass
    "points_65536":  ((1, _S2, 1, 32, 16384, 4), ["acc=wide"]),                          # Lq*P = kWideMaxStep: one attempt covers all points
    "points_65540":  ((1, _S2, 1, 32, 16385, 4), ["acc=wide"]),                          # two attempts (16-bit list entries)
    "roleB_384":     ((2, _C2, 8, 32, 3060, 4), ["bwd=fused_lds(", "roleB=384", "roleA=128"]),   # role A in the slots role B leaves
    "roleB_448":     ((2, _C2, 8, 32, 3300, 4), ["bwd=fused_lds(", "roleB=448", "roleA=512"]),   # ... behind several rounds of role B
}


@pytest.mark.parametrize("name", list(STRADDLE))
def test_either_side_of_every_dispatcher_threshold(native, oracle, name):
    """Each geometry sits just below or just above one of the launch-plan thresholds; the plan the library reports
    (msda_describe_plan) must be the branch named here, and forward + backward must match the C oracle on it."""
    case, must = STRADDLE[name]
    N, shapes, M, D, Lq, P = case
    S = sum(h * w for h, w in shapes)
    plan = native.describe_plan(N, S, M, D, len(shapes), Lq, P)
    for sub in must:
        assert sub in plan, (sub, plan)
    z = make_case(21, *case)
    out, gv, gl, ga = run_hip(z, torch.float32)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(out, oracle.forward(*args)) < 5e-6
    # (thousands of taps per pixel on these tiny maps: fp32 summation order shows at 5e-5 of max)
    assert rel_err(gv, r_gv) < 5e-5
    assert rel_err(ga, r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(gl[keep], r_gl[keep]) < 2e-5
