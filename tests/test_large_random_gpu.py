"""Seeded random geometries in the LARGE regime (N*M*Lq >= 32768 items: the LDS-stage forward / role A, the kept-taps and
single-pass role B plans, the forward's range masks, the dense coarse levels) against the C oracle — the random geometries of
test_parity_gpu.py all stay below that size.  Default and deterministic backward, uniform and clustered sampling locations."""
import numpy as np
import pytest
import torch

from conftest import near_boundary_mask, rel_err

pytestmark = pytest.mark.gpu


def _large_geometries(count, seed, pow2=False):
    """pow2: L*P and P powers of two, M a divisor of 64 — what the fused prologue takes (msda_prologue_supported, include/msda.h)."""
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < count:
        L = int(rng.choice([1, 2, 4, 4])) if pow2 else int(rng.randint(1, 6))
        P = int(rng.choice([1, 2, 4, 4])) if pow2 else int(rng.choice([1, 2, 3, 4]))
        h, w = int(rng.randint(6, 57)), int(rng.randint(6, 57))
        if rng.rand() < 0.7:                                             # a pyramid (odd sizes halve unevenly on purpose)
            shapes = [(max(1, -(-h >> k)), max(1, -(-w >> k))) for k in range(L)]
        else:
            shapes = [(int(rng.randint(1, 41)), int(rng.randint(1, 41))) for _ in range(L)]
        M = int(rng.choice([4, 8, 8])) if pow2 else int(rng.choice([4, 8, 8, 8, 6]))      # (the prologue: M divides 64)
        N = int(rng.randint(1, 7))
        lq_min = -(-32768 // (N * M))
        Lq = int(rng.randint(lq_min, max(lq_min + 1, min(4 * lq_min, 3500))))
        if N * M * Lq * L * P > 3_000_000:                               # keeps the oracle to about a second per case
            continue
        spread = float(rng.choice([0.0, 0.0, 0.05, 0.3]))                # 0: uniform over [-0.25, 1.25]; else clustered taps
        cases.append((N, shapes, M, Lq, P, spread))
    return cases


def _make(seed, N, shapes, M, Lq, P, spread):
    g = torch.Generator().manual_seed(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    L, S = len(shapes), int(shapes.prod(1).sum())
    lsi = np.concatenate(([0], np.cumsum(shapes.prod(1))[:-1])).astype(np.int64)
    value = torch.randn(N, S, M, 32, generator=g)
    if spread == 0.0:
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.5 - 0.25
    else:                                                               # reference points + small offsets, as the model makes them
        ref = torch.rand(N, Lq, 1, 1, 1, 2, generator=g)
        loc = ref + torch.randn(N, Lq, M, L, P, 2, generator=g) * spread
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = attn / attn.sum((-1, -2), keepdim=True)
    go = torch.randn(N, Lq, M * 32, generator=g)
    return dict(value=value.numpy(), shapes=shapes, level_start=lsi, loc=loc.numpy(), attn=attn.numpy(), grad_out=go.numpy())


def _run(z, deterministic):
    from uvhand_amd.functions import MSDeformAttnFunction
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    v, l, a = dv(z["value"]).requires_grad_(True), dv(z["loc"]).requires_grad_(True), dv(z["attn"]).requires_grad_(True)
    torch.use_deterministic_algorithms(deterministic)
    try:
        out = MSDeformAttnFunction.apply(v, dv(z["shapes"]), dv(z["level_start"]), l, a, 64)
        out.backward(dv(z["grad_out"]))
        torch.cuda.synchronize()
    finally:
        torch.use_deterministic_algorithms(False)
    return [t.detach().cpu().numpy() for t in (out, v.grad, l.grad, a.grad)]


@pytest.mark.parametrize("idx,case", list(enumerate(_large_geometries(28, 555))))
def test_large_random_geometries_match_c_oracle(oracle, idx, case):
    from uvhand_amd import _native
    _native.load()
    z = _make(900 + idx, *case)
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_out = oracle.forward(*args)
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    keep = ~near_boundary_mask(z, tol=1e-5)
    for det in (False, True):
        out, gv, gl, ga = _run(z, det)
        what = (case, "deterministic" if det else "default")
        assert rel_err(out, r_out) < 5e-6, what
        assert rel_err(gv, r_gv) < 2e-5, what
        assert rel_err(ga, r_ga) < 2e-5, what
        if keep.any():
            assert rel_err(gl[keep], r_gl[keep]) < 2e-5, what


def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize("idx,case", list(enumerate(_large_geometries(12, 8080))))
def test_large_random_geometries_bf16_rows(oracle, idx, case):
    """bf16 rows (both grad_value types) on large random geometries, the backward fed from the forward's table as under autograd:
    inputs pre-rounded to bf16, so only the final rounding (4e-3 of max) and fp32 summation order (2e-5) separate it from the oracle."""
    from uvhand_amd import _native
    _native.load()
    z = _make(1700 + idx, *case)
    z["value"], z["grad_out"] = _bf16_round(z["value"]), _bf16_round(z["grad_out"])
    args = [z["value"], z["shapes"], z["level_start"], z["loc"], z["attn"]]
    r_out = oracle.forward(*args)
    r_gv, r_gl, r_ga = oracle.backward(z["grad_out"], *args)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    v16, go16 = dv(z["value"]).to(torch.bfloat16), dv(z["grad_out"]).to(torch.bfloat16)
    s, i, l, a = dv(z["shapes"]), dv(z["level_start"]), dv(z["loc"]), dv(z["attn"])
    out, table = _native.ms_deform_attn_forward(v16, s, i, l, a, 64, with_table=True)
    assert rel_err(out.float().cpu().numpy(), r_out) < 4e-3, case
    keep = ~near_boundary_mask(z, tol=1e-5)
    for gv32, tol in ((False, 4e-3), (True, 2e-5)):
        gv, gl, ga = _native.ms_deform_attn_backward(v16, s, i, l, a, go16, 64, fp32_grad_value=gv32, table=table)
        assert rel_err(gv.float().cpu().numpy(), r_gv) < tol, (case, gv32)
        assert rel_err(ga.cpu().numpy(), r_ga) < 2e-5, (case, gv32)
        if keep.any():
            assert rel_err(gl.cpu().numpy()[keep], r_gl[keep]) < 2e-5, (case, gv32)


@pytest.mark.parametrize("idx,case", list(enumerate(_large_geometries(12, 31337, pow2=True))))
def test_large_random_geometries_through_the_fused_prologue(oracle, idx, case):
    """What the MODULE runs (reference points + raw offsets + logits in, their gradients out; the forward's table handed to the
    backward): the oracle on the locations / weights the prologue forward returns, the chain rule of
    models/ops/modules/ms_deform_attn.py:101-108 applied to its gradients on the host (test_lds_prologue_gpu._host_chain_rule)."""
    from test_lds_prologue_gpu import _host_chain_rule
    from uvhand_amd import _native
    _native.load()
    N, shapes, M, Lq, P, spread = case
    assert _native.prologue_geometry_supported(N, sum(h * w for h, w in shapes), M, 32, len(shapes), Lq, P), case
    g = torch.Generator().manual_seed(4000 + idx)
    sh = torch.tensor(shapes, dtype=torch.long)
    L, S = len(shapes), sum(h * w for h, w in shapes)
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    value = torch.rand(N, S, M, 32, generator=g) - 0.5
    ref = torch.rand(N, Lq, L, 2, generator=g) * 1.2 - 0.1
    off = torch.randn(N, Lq, M, L, P, 2, generator=g) * (2.0 if spread == 0.0 else 40.0 * spread)
    logits = torch.randn(N, Lq, M, L * P, generator=g)
    go = torch.randn(N, Lq, M * 32, generator=g)
    dv = lambda t: t.cuda()
    out, loc, attn, table = _native.ms_deform_attn_forward_prologue(dv(value), dv(sh), dv(lsi), dv(ref), dv(off), dv(logits), 64,
                                                                    with_table=True)
    gv, goff, glog, gref = _native.ms_deform_attn_backward_prologue(dv(value), dv(sh), dv(lsi), loc, attn, dv(go), table=table)
    torch.cuda.synchronize()
    shn = sh.numpy()
    args = [value.numpy(), shn, lsi.numpy(), loc.cpu().numpy(), attn.cpu().numpy()]
    assert rel_err(out.cpu().numpy(), oracle.forward(*args)) < 5e-6, case
    r_gv, r_gl, r_ga = oracle.backward(go.numpy(), *args)
    assert rel_err(gv.cpu().numpy(), r_gv) < 2e-5, case
    r_off, r_log, r_ref = _host_chain_rule(shn, args[3], args[4], r_gl, r_ga)
    assert rel_err(glog.cpu().numpy(), r_log.reshape(glog.shape)) < 2e-5, case
    pix = args[3].astype(np.float64) * np.stack([shn[:, 1], shn[:, 0]], -1).astype(np.float64)[None, None, None, :, None, :] - 0.5
    near = (np.abs(pix - np.round(pix)) < 1e-5).any(-1)
    keep, keep_cell = ~near, ~near.any(axis=(2, 4))
    assert rel_err(goff.cpu().numpy()[keep], r_off[keep]) < 2e-5, case
    assert rel_err(gref.cpu().numpy()[keep_cell], r_ref[keep_cell]) < 2e-5, case
