"""Coarse levels on the matrix cores (uvhand_amd/csrc/msda_d32_dense.h) and the padded record layout of the kept-taps pass
(gather_pad8, msda_d32_value.h): parity with the C oracle on geometries that sit on the edges of both — levels of 16 / 17 /
32 / 33 / 64 / 65 pixels with one and with two workgroups per level, query counts around the 512-query chunk, P != 4,
1-pixel-wide levels, a dense level 0, the deterministic flag and bf16 rows.  All need the GPU (`-m gpu`)."""
import numpy as np
import pytest
import torch

from conftest import near_boundary_mask, rel_err
from test_parity_gpu import dev, make_case, run_hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from uvhand_amd import _native
    _native.load()
    assert torch.cuda.is_available()
    return _native


DENSE = {
    # name: ((N, shapes, M, D, Lq, P), substrings the launch plan must contain)
    # kept-taps pass (Lq*P > 1536), two workgroups per level: levels of <= 64 pixels dense, two 32-pixel passes above 32
    "w2_64_32_16":     ((4, [(20, 20), (8, 8), (4, 8), (4, 4)], 8, 32, 1100, 4), ["bwd=fused_lds(acc=wide,W=2", "dense_px=64"]),
    "w2_65_33_17":     ((4, [(20, 20), (5, 13), (3, 11), (1, 17)], 8, 32, 1100, 4), ["bwd=fused_lds(acc=wide,W=2", "dense_px=64"]),
    "w2_chunk_edges":  ((8, [(18, 18), (7, 7), (1, 1)], 8, 32, 513, 4), ["bwd=fused_lds(acc=wide,W=2", "dense_px=64"]),     # 512 + 1 queries
    "w2_lq_1023":      ((6, [(16, 16), (6, 6), (2, 3)], 8, 32, 1023, 4), ["bwd=fused_lds(acc=wide", "dense_px="]),
    "w2_p8":           ((8, [(16, 16), (4, 4)], 8, 32, 520, 8), ["bwd=fused_lds(acc=wide", "dense_px="]),                 # P = 8: the scalar point loop
    "w2_p2_l5":        ((8, [(14, 14), (7, 7), (4, 4), (2, 2), (1, 1)], 8, 32, 1100, 2), ["bwd=fused_lds(acc=wide", "dense_px="]),
    # single pass (Lq*P <= 1536), one workgroup per level: levels of <= 32 pixels dense, 33..64 through the sort
    "w1_49_32_16":     ((16, [(14, 14), (7, 7), (2, 16), (4, 4)], 8, 32, 300, 4), ["bwd=fused_lds(acc=single,W=1", "dense_px=32"]),
    "w1_dense_level0": ((32, [(4, 8), (4, 4)], 8, 32, 256, 4), ["bwd=fused_lds(acc=single", "dense_px="]),                # level 0 dense: it also zeroes nothing else
    "w1_p3":           ((16, [(12, 12), (5, 6), (16, 1)], 8, 32, 301, 3), ["bwd=fused_lds(acc=single", "dense_px="]),
    "w1_m5":           ((24, [(10, 10), (4, 4)], 5, 32, 333, 4), ["bwd=fused_lds(", "dense_px="]),                        # M = 5: unaligned head stride
}


def _check(native, oracle, name, dtype=torch.float32, deterministic=False):
    case, must = DENSE[name]
    N, shapes, M, D, Lq, P = case
    S = sum(h * w for h, w in shapes)
    plan = native.describe_plan(N, S, M, D, len(shapes), Lq, P, deterministic=deterministic)
    for sub in must:
        assert sub in plan, (sub, plan)
    z = make_case(33, *case)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    r_out = oracle.forward(*args)
    if deterministic:
        import os
        os.environ["MSDA_DETERMINISTIC"] = "1"
    try:
        out, gv, gl, ga = run_hip(z, dtype)
    finally:
        if deterministic:
            os.environ.pop("MSDA_DETERMINISTIC", None)
    return (out, gv, gl, ga), (r_out, r_gv, r_gl, r_ga), z


@pytest.mark.parametrize("name", list(DENSE))
def test_dense_levels_against_the_c_oracle(native, oracle, name):
    (out, gv, gl, ga), (r_out, r_gv, r_gl, r_ga), z = _check(native, oracle, name)
    assert rel_err(out, r_out) < 5e-6
    # per level too: a wrong tile of a 16-pixel level would hide behind the 400-pixel level's rows in a whole-tensor norm
    shapes, ls = z["shapes"], z["level_start"]
    for l in range(len(shapes)):
        a, b = int(ls[l]), int(ls[l] + shapes[l].prod())
        assert rel_err(gv[:, a:b], r_gv[:, a:b]) < 5e-5, "level %d" % l
    assert rel_err(ga, r_ga) < 2e-5
    keep = ~near_boundary_mask(z, tol=1e-5)
    assert rel_err(gl[keep], r_gl[keep]) < 2e-5


@pytest.mark.parametrize("name", ["w2_65_33_17", "w1_49_32_16"])
def test_dense_levels_with_the_deterministic_flag(native, oracle, name):
    """The dense body's summation order is fixed, so the deterministic launches take it too: same values as the oracle, and
    twice the same bits."""
    (out, gv, gl, ga), (r_out, r_gv, r_gl, r_ga), z = _check(native, oracle, name, deterministic=True)
    assert rel_err(gv, r_gv) < 5e-5
    (_, gv2, _, _), _, _ = _check(native, oracle, name, deterministic=True)
    assert np.array_equal(gv, gv2)


@pytest.mark.parametrize("name", ["w2_64_32_16", "w1_49_32_16"])
def test_dense_levels_every_run_the_same_bits_on_the_dense_rows(native, oracle, name):
    """Without the flag the sort + gather levels may differ in the last bit from run to run; the dense levels may not."""
    (_, gv, _, _), _, z = _check(native, oracle, name)
    (_, gv2, _, _), _, _ = _check(native, oracle, name)
    shapes, ls = z["shapes"], z["level_start"]
    limit = 64 if "w2" in name else 32
    for l in range(len(shapes)):
        if int(shapes[l].prod()) <= limit:
            a, b = int(ls[l]), int(ls[l] + shapes[l].prod())
            assert np.array_equal(gv[:, a:b], gv2[:, a:b]), "level %d" % l


@pytest.mark.parametrize("name", ["w2_64_32_16", "w1_49_32_16"])
def test_dense_levels_bf16_rows(native, oracle, name):
    """bf16 value / grad_out rows (fp32 arithmetic, one rounding at the store): against the oracle run on the bf16-rounded
    inputs, at bf16 resolution."""
    case, _ = DENSE[name]
    z = make_case(34, *case)
    rnd = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    z["value"], z["grad_out"] = rnd(z["value"]), rnd(z["grad_out"])
    z["loc"], z["attn"] = rnd(z["loc"]), rnd(z["attn"])
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_gv, _, _ = oracle.backward(z["grad_out"], *args)
    from uvhand_amd.functions import MSDeformAttnFunction
    v = dev(z["value"], torch.bfloat16).requires_grad_(True)
    out = MSDeformAttnFunction.apply(v, dev(z["shapes"]), dev(z["level_start"]), dev(z["loc"], torch.bfloat16),
                                     dev(z["attn"], torch.bfloat16), 64)
    out.backward(dev(z["grad_out"], torch.bfloat16))
    torch.cuda.synchronize()
    assert rel_err(v.grad.float().cpu().numpy(), r_gv) < 1e-2


def _rows_hit_by(z, b, q, m):
    """Pixel rows (indices into S) of batch element b that a VALID tap of query q, head m lands on — the rows the reference's
    atomicAdd touches for that query (ms_deform_im2col_cuda.cuh:56-78 guards, :125-152 adds)."""
    rows = set()
    for l, (H, Wd) in enumerate(z["shapes"]):
        H, Wd = int(H), int(Wd)
        for p in range(z["loc"].shape[4]):
            x, y = np.float32(z["loc"][b, q, m, l, p, 0]), np.float32(z["loc"][b, q, m, l, p, 1])
            h_im, w_im = y * np.float32(H) - np.float32(0.5), x * np.float32(Wd) - np.float32(0.5)
            if not (h_im > -1 and w_im > -1 and h_im < H and w_im < Wd):
                continue
            h0, w0 = int(np.floor(h_im)), int(np.floor(w_im))
            for hh, ww in ((h0, w0), (h0, w0 + 1), (h0 + 1, w0), (h0 + 1, w0 + 1)):
                if 0 <= hh < H and 0 <= ww < Wd:
                    rows.add(int(z["level_start"][l]) + hh * Wd + ww)
    return rows


@pytest.mark.parametrize("name", ["w2_64_32_16", "w1_49_32_16"])
def test_which_rows_a_non_finite_grad_out_row_poisons(native, name):
    """one Inf in grad_out[b, q, m-th head's channels].  Reference semantics (atomicAdd of weight * grad_out
    per tap, ms_deform_im2col_cuda.cuh:125-152): exactly the rows that query's taps land on become non-finite.
    * MSDA_FLAG_EXACT_NONFINITE (uvhand_amd.set_exact_nonfinite): exactly those rows, on every level;
    * default: the same on the sort + gather levels; on a DENSE level (matrix-core product: a zero weight still multiplies the
      row) every row of that level of (b, m) — a superset, confined to that (batch, head, level);
    nothing outside (b, m) is touched either way, and grad_sampling_loc / grad_attn_weight are non-finite for that one item only."""
    case, _ = DENSE[name]
    N, shapes, M, D, Lq, P = case
    S = sum(h * w for h, w in shapes)
    z = make_case(35, *case)
    b, q, m = 1, 7, 3
    z["grad_out"].reshape(N, Lq, M, D)[b, q, m, 5] = np.inf
    v, go = dev(z["value"]), dev(z["grad_out"])
    s, i, l, a = dev(z["shapes"]), dev(z["level_start"]), dev(z["loc"]), dev(z["attn"])
    want = _rows_hit_by(z, b, q, m)
    assert want, "the chosen query has no valid tap"
    limit = 64 if "w2" in name else 32
    dense_rows = set()
    for lv, (H, Wd) in enumerate(shapes):
        if H * Wd <= limit and any(int(z["level_start"][lv]) <= r < int(z["level_start"][lv]) + H * Wd for r in want):
            dense_rows |= set(range(int(z["level_start"][lv]), int(z["level_start"][lv]) + H * Wd))
    assert dense_rows, "no dense level is hit"
    assert "dense_px=0" in native.describe_plan(N, S, M, D, len(shapes), Lq, P, exact_nonfinite=True)
    for exact in (False, True):
        native.set_exact_nonfinite(exact)
        try:
            gv, gl, ga = native.ms_deform_attn_backward(v, s, i, l, a, go, 64)
        finally:
            native.set_exact_nonfinite(False)
        bad = ~torch.isfinite(gv).all(-1)                                   # [N, S, M]
        got = set(torch.nonzero(bad[b, :, m]).flatten().tolist())
        assert got == (want if exact else want | dense_rows), (exact, sorted(got ^ want)[:10])
        bad[b, :, m] = False
        assert not bad.any(), "rows outside the (batch, head) pair touched"
        bad_items = ~(torch.isfinite(gl).all(-1).all(-1).all(-1) & torch.isfinite(ga).all(-1).all(-1))      # [N, Lq, M]
        assert set(map(tuple, torch.nonzero(bad_items).tolist())) == {(b, q, m)}
