"""The deterministic backward (MSDA_FLAG_DETERMINISTIC, include/msda.h; the DET form of role B in
uvhand_amd/csrc/msda_d32_value.h — per-wavefront counters — and the destination-major kernel of msda_generic.hip): parity
with the C oracle on the same seeded geometries as the default kernels, bitwise reproducibility run to run — which the
reference's atomicAdd scatter (ms_deform_im2col_cuda.cuh:125-152) and the default kernels do not have — for fp32 and bf16
rows, through the fused-prologue entry point and the kept-taps pass with its fall-back chunks, and memory safety on shapes
that are inconsistent with S."""
import numpy as np
import pytest
import torch

from conftest import near_boundary_mask, rel_err
from test_parity_gpu import ORACLE_CASES, _random_geometries, dev, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from uvhand_amd import _native
    _native.load()
    return _native


def _backward(native, z, deterministic=True, dtype=torch.float32):
    t = {k: dev(z[k], dtype) for k in ("value", "loc", "attn", "grad_out")}
    gv, gl, ga = native.ms_deform_attn_backward(t["value"], dev(z["shapes"]), dev(z["level_start"]), t["loc"], t["attn"],
                                                t["grad_out"], 64, deterministic=deterministic)
    torch.cuda.synchronize()
    return gv, gl, ga


def _check(z, got, oracle, tol=2e-5):
    gv, gl, ga = (t.float().cpu().numpy() for t in got)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    assert rel_err(gv, r_gv) < tol
    assert rel_err(ga, r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    if keep.any():
        assert rel_err(gl[keep], r_gl[keep]) < 2e-5


D32_CASES = [k for k, c in ORACLE_CASES.items() if c[3] == 32]
EXTRA = {
    # the geometries round 1's review singled out: many (batch, head, level) triples on one large level, and a long one
    "one_big_level": (32, [(40, 48)], 8, 32, 384, 4),
    "long_level":    (2, [(300, 3), (1, 200)], 4, 32, 77, 2),                  # tiles of 85 x 3 and 1 x 16 pixels
    "encoder_like":  (1, [(20, 20), (10, 10), (5, 5)], 8, 32, 525, 4),          # Lq*P > one batch per tile
    "chunked":       (1, [(16, 16), (8, 8)], 2, 32, 4000, 4),                   # few pairs, many points: query chunks + slabs
}


@pytest.mark.parametrize("name", D32_CASES + list(EXTRA))
def test_deterministic_matches_c_oracle(native, oracle, name):
    z = make_case(3, *(ORACLE_CASES[name] if name in ORACLE_CASES else EXTRA[name]))
    _check(z, _backward(native, z), oracle)


@pytest.mark.parametrize("idx,case", list(enumerate(_random_geometries(24, 777))))
def test_deterministic_random_geometries(native, oracle, idx, case):
    z = make_case(500 + idx, *case)
    _check(z, _backward(native, z), oracle)


def test_deterministic_mode_needs_no_scratch(native, oracle):
    """The per-wavefront counters live in LDS: no workspace for the flag on the plain entry points, whatever the geometry, and
    the same result with and without a buffer."""
    case = EXTRA["chunked"]
    N, shapes, M, D, Lq, P = case
    S = sum(h * w for h, w in shapes)
    lib = native.load()
    assert lib.msda_backward_workspace_bytes(N, S, M, D, len(shapes), Lq, P, native.FLAG_DETERMINISTIC) == 0
    assert lib.msda_backward_workspace_bytes(2, 3060, 8, 32, 4, 3060, 4, native.FLAG_DETERMINISTIC) == 0
    z = make_case(9, *case)
    through_python = _backward(native, z)
    _check(z, through_python, oracle)
    import ctypes
    t = {k: dev(z[k]) for k in ("value", "loc", "attn", "grad_out")}
    sh, ls = dev(z["shapes"]), dev(z["level_start"])
    gv, gl, ga = torch.empty_like(t["value"]), torch.empty_like(t["loc"]), torch.empty_like(t["attn"])
    fn = lib.msda_backward_ws_f32
    fn.argtypes = native._BWD_WS_ARGTYPES
    fn.restype = ctypes.c_int
    scratch = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    rc = fn(t["grad_out"].data_ptr(), t["value"].data_ptr(), sh.data_ptr(), ls.data_ptr(), t["loc"].data_ptr(),
            t["attn"].data_ptr(), N, S, M, D, len(shapes), Lq, P, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), scratch.data_ptr(),
            scratch.numel(), native.FLAG_DETERMINISTIC, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip((gv, gl, ga), through_python))


@pytest.mark.parametrize("name,spread", [("cfg2_decoder", None), ("pile_up", 0.02), ("chunked", None)])
def test_grad_value_is_bitwise_reproducible(native, name, spread):
    if name == "chunked":
        z = make_case(5, *EXTRA["chunked"])
    else:
        z = make_case(11, 2, [(48, 48), (24, 24), (12, 12), (6, 6)], 8, 32, 300, 4)
        if spread is not None:
            g = torch.Generator().manual_seed(12)
            z["loc"] = (torch.tensor([0.37, 0.61]) + (torch.rand(z["loc"].shape, generator=g) - 0.5) * spread).numpy().astype(np.float32)
    first = _backward(native, z)
    # other work in between perturbs the timing of the workgroups
    noise = torch.randn(1 << 20, device="cuda")
    for i in range(10):
        if i % 3 == 0:
            noise = noise * 1.0001 + 1.0
        again = _backward(native, z)
        for a, b in zip(first, again):
            assert torch.equal(a, b), "run %d differs" % i


def test_deterministic_bf16_rows(native, oracle):
    z = make_case(4, *ORACLE_CASES["many_queries"])
    for k in ("value", "grad_out"):
        z[k] = torch.from_numpy(z[k]).to(torch.bfloat16).float().numpy()
    t = {k: dev(z[k]) for k in ("value", "loc", "attn", "grad_out")}
    args = (t["value"].to(torch.bfloat16), dev(z["shapes"]), dev(z["level_start"]), t["loc"], t["attn"],
            t["grad_out"].to(torch.bfloat16), 64)
    for fp32_gv, tol in ((False, 4e-3), (True, 2e-5)):
        got = native.ms_deform_attn_backward(*args, fp32_grad_value=fp32_gv, deterministic=True)
        assert got[0].dtype == (torch.float32 if fp32_gv else torch.bfloat16)
        _check(z, got, oracle, tol=tol)
        again = native.ms_deform_attn_backward(*args, fp32_grad_value=fp32_gv, deterministic=True)
        assert all(torch.equal(a, b) for a, b in zip(got, again))


def test_function_and_module_follow_torch_deterministic_switch(native, oracle):
    from uvhand_amd.functions import MSDeformAttnFunction
    z = make_case(21, *ORACLE_CASES["model_small"])
    grads = []
    prev = torch.are_deterministic_algorithms_enabled()
    try:
        torch.use_deterministic_algorithms(True, warn_only=True)
        assert native.deterministic_requested()
        for _ in range(3):
            v, l, a = (dev(z[k]).requires_grad_(True) for k in ("value", "loc", "attn"))
            MSDeformAttnFunction.apply(v, dev(z["shapes"]), dev(z["level_start"]), l, a, 64).backward(dev(z["grad_out"]))
            grads.append(v.grad.clone())
    finally:
        torch.use_deterministic_algorithms(prev)
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    _check(z, (grads[0], l.grad, a.grad), oracle)


def test_inconsistent_shapes_are_memory_safe(native):
    """A level that does not fit in S contributes nothing, pixels no level covers get zeros, nothing faults."""
    z = make_case(2, 1, [(6, 6), (3, 3)], 8, 32, 40, 4)
    S = z["value"].shape[1]
    bad = dict(z)
    bad["level_start"] = np.asarray([0, S - 4], dtype=np.int64)               # second level would run past S
    gv, gl, ga = _backward(native, bad)
    assert torch.isfinite(gv).all() and torch.isfinite(gl).all() and torch.isfinite(ga).all()
    assert torch.count_nonzero(gv[:, 36:]) == 0                                # rows of the dropped level: zeros
    short = dict(z)
    short["shapes"] = np.asarray([(6, 6), (2, 2)], dtype=np.int64)             # covers S - 5 pixels only
    gv2, _, _ = _backward(native, short)
    assert torch.count_nonzero(gv2[:, 40:]) == 0


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("Lq", [40, 700])                                     # single pass / kept-taps pass of the default kernels
def test_rows_no_level_covers_are_zero_wherever_they_lie(native, deterministic, Lq):
    """Gaps BEFORE and BETWEEN the levels (level_start[0] > 0, a middle level that does not fit) — not only a trailing one —
    get zeros from both role-B generations; grad_value comes from torch.empty, so anything unwritten would show."""
    z = make_case(5, 1, [(6, 6), (3, 3), (2, 2)], 8, 32, Lq, 4)
    S = z["value"].shape[1]                                                    # 49
    # [3, 39, 48]: 3 uncovered rows in front, the last level (4 pixels at 48) runs past S -> dropped, row 48 uncovered;
    # [0, 47, 45]: the MIDDLE level (9 pixels at 47) does not fit -> rows 36..44 uncovered, the last level sits at 45..48
    for lsi in ([3, 39, 48], [0, S - 2, 45]):
        bad = dict(z)
        bad["level_start"] = np.asarray(lsi, dtype=np.int64)
        for _ in range(2):                                                     # the allocator hands back dirty blocks
            torch.full((2, S, 8, 32), float("nan"), device="cuda")
        gv, gl, ga = _backward(native, bad, deterministic=deterministic)
        assert torch.isfinite(gv).all() and torch.isfinite(gl).all() and torch.isfinite(ga).all()
        covered = torch.zeros(S, dtype=torch.bool)
        for (h, w), st in zip([(6, 6), (3, 3), (2, 2)], lsi):
            if st + h * w <= S:
                covered[st:st + h * w] = True
        assert torch.count_nonzero(gv[:, ~covered.cuda()]) == 0
        assert gv[:, covered.cuda()].abs().sum() > 0


@pytest.mark.parametrize("M,Lq,shapes", [(8, 37, [(12, 12), (6, 6), (3, 3), (2, 2)]),          # one launch (roles fused)
                                         (8, 700, [(16, 16), (8, 8), (4, 4), (2, 2)]),          # Lq*P > 2048: role A as its own launch
                                         (16, 600, [(10, 10), (5, 5)]),                           # 16 heads: whole queries per workgroup
                                         (2, 2100, [(16, 16), (8, 8), (4, 4), (2, 2)])])          # few pairs: query chunks + slabs
def test_deterministic_fused_prologue_backward(native, M, Lq, shapes):
    """msda_backward_prologue_ws_f32 with MSDA_FLAG_DETERMINISTIC: the raw-tensor gradients equal the default kernels'
    (same role A arithmetic: bit for bit) and grad_value matches within summation order and is reproducible."""
    g = torch.Generator().manual_seed(M * 1000 + Lq)
    N, P, L = 2, 4, len(shapes)
    S = sum(h * w for h, w in shapes)
    sh = torch.tensor(shapes, dtype=torch.long).cuda()
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    if not native.prologue_geometry_supported(N, S, M, 32, L, Lq, P):
        pytest.skip("geometry outside the fused prologue")
    value = (torch.rand(N, S, M, 32, generator=g) - 0.5).cuda()
    ref = (torch.rand(N, Lq, L, 2, generator=g) * 1.4 - 0.2).cuda()
    off = (torch.randn(N, Lq, M, L, P, 2, generator=g) * 2.0).cuda()
    logits = torch.randn(N, Lq, M, L * P, generator=g).cuda()
    go = torch.randn(N, Lq, M * 32, generator=g).cuda()
    out, loc, attn = native.ms_deform_attn_forward_prologue(value, sh, lsi, ref, off, logits, 64)
    base = native.ms_deform_attn_backward_prologue(value, sh, lsi, loc, attn, go, deterministic=False)
    det1 = native.ms_deform_attn_backward_prologue(value, sh, lsi, loc, attn, go, deterministic=True)
    det2 = native.ms_deform_attn_backward_prologue(value, sh, lsi, loc, attn, go, deterministic=True)
    torch.cuda.synchronize()
    for a, b in zip(det1, det2):
        assert torch.equal(a, b)
    assert rel_err(det1[0].cpu().numpy(), base[0].cpu().numpy()) < 2e-5            # grad_value
    for k in (1, 2, 3):                                                             # offsets, logits, reference points
        assert torch.equal(det1[k], base[k])


@pytest.mark.parametrize("name", ["cfg2_decoder", "cfg2_encoder", "cfg4_decoder", "cfg4_encoder"])
def test_deterministic_full_size_shapes(native, name):
    """BASELINE's full sizes: the adjoint identity <out, go> = <value, grad_value> (the op is linear in value), agreement
    with the default kernels, and reproducibility.  cfg-2 encoder is the chunked case (query chunks + slab reduce), the
    cfg-4 shapes run role B and role A as two launches."""
    from test_parity_gpu import FULL
    z = make_case(0, *FULL[name], lo=-0.1, hi=1.1)
    t = {k: dev(z[k]) for k in ("value", "loc", "attn", "grad_out")}
    sh, ls = dev(z["shapes"]), dev(z["level_start"])
    out = native.ms_deform_attn_forward(t["value"], sh, ls, t["loc"], t["attn"], 64)
    base = native.ms_deform_attn_backward(t["value"], sh, ls, t["loc"], t["attn"], t["grad_out"], 64, deterministic=False)
    det = native.ms_deform_attn_backward(t["value"], sh, ls, t["loc"], t["attn"], t["grad_out"], 64, deterministic=True)
    det2 = native.ms_deform_attn_backward(t["value"], sh, ls, t["loc"], t["attn"], t["grad_out"], 64, deterministic=True)
    torch.cuda.synchronize()
    dot = (out.double() * t["grad_out"].double()).sum().item()
    assert abs(dot - (t["value"].double() * det[0].double()).sum().item()) < 1e-5 * abs(dot)
    assert rel_err(det[0].cpu().numpy(), base[0].cpu().numpy()) < 2e-5
    assert torch.equal(det[1], base[1]) and torch.equal(det[2], base[2])          # role A is the same arithmetic
    assert all(torch.equal(a, b) for a, b in zip(det, det2))


@pytest.mark.parametrize("name,dtype", [("testpy_D30", torch.float32), ("wide_D64", torch.float32), ("testpy_D30", torch.float64),
                                        ("cfg1_like", torch.float32)])
def test_deterministic_flag_reaches_the_generic_family(native, oracle, name, dtype):
    """Outside the D = 32 family (any D, fp64, a forced path) MSDA_FLAG_DETERMINISTIC selects the destination-major kernel of
    msda_generic.hip: grad_value equals the oracle's, identically on every run; grad_loc / grad_attn are the default kernel's."""
    cases = {"testpy_D30": (3, [(6, 4), (3, 2)], 2, 30, 7, 2), "wide_D64": (2, [(9, 7), (4, 5), (2, 2)], 3, 64, 50, 4),
             "cfg1_like": (1, [(16, 16), (8, 8)], 8, 32, 60, 4)}
    z = make_case(11, *cases[name])
    force = name == "cfg1_like"                              # a D = 32 geometry pushed onto the generic kernels
    if force:
        native.force_path(native.PATH_GENERIC)
    try:
        runs = [_backward(native, z, deterministic=True, dtype=dtype) for _ in range(3)]
        base = _backward(native, z, deterministic=False, dtype=dtype)
    finally:
        native.force_path(-1)
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], r))
    assert torch.equal(runs[0][1], base[1]) and torch.equal(runs[0][2], base[2])
    _check(z, runs[0], oracle)                           # (the oracle runs in the inputs' precision: float32)


@pytest.mark.parametrize("spread", [None, 0.02], ids=["uniform", "pile_up"])
def test_deterministic_small_problem_body_with_and_without_the_forward_table(native, oracle, spread):
    """Round 5: small problems (every workgroup resident: the 300-query decoder shape) keep the short fixed-capacity sort under
    the deterministic flag — per-wavefront counters, ranks in list order — instead of the general prefix-sum body.  The
    record order is a pure function of the inputs, so the backward from the forward's point table and the backward from a scan
    of sampling_loc give the SAME BITS (the default kernels agree only up to summation order), run after run; piled-up
    locations overflow the fixed segments and take the general deterministic body, with the same guarantees."""
    N, shapes, M, D, Lq, P = 2, [(48, 48), (24, 24), (12, 12), (6, 6)], 8, 32, 300, 4
    S = sum(h * w for h, w in shapes)
    assert "fixed,det" in native.describe_plan(N, S, M, D, len(shapes), Lq, P, deterministic=True)
    z = make_case(21, N, shapes, M, D, Lq, P)
    if spread is not None:
        g = torch.Generator().manual_seed(13)
        z["loc"] = (torch.tensor([0.37, 0.61]) + (torch.rand(z["loc"].shape, generator=g) - 0.5) * spread).numpy().astype(np.float32)
    t = {k: dev(z[k]) for k in ("value", "loc", "attn", "grad_out")}
    sh, ls = dev(z["shapes"]), dev(z["level_start"])
    _, table = native.ms_deform_attn_forward(t["value"], sh, ls, t["loc"], t["attn"], 64, with_table=True)
    assert table is not None
    scan = native.ms_deform_attn_backward(t["value"], sh, ls, t["loc"], t["attn"], t["grad_out"], 64, deterministic=True)
    for i in range(4):
        tab = native.ms_deform_attn_backward(t["value"], sh, ls, t["loc"], t["attn"], t["grad_out"], 64, deterministic=True, table=table)
        assert all(torch.equal(a, b) for a, b in zip(scan, tab)), "run %d" % i
    _check(z, scan, oracle)
