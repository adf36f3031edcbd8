"""Parity of the kernels the MODULE runs at training geometry: the fused prologue on the LDS-stage forward / role A
(msda_d32_lds.h: fwd_d32_lds_kernel<., true, .>, bwd_fused_lds_d32_kernel<., ., true, ...>) and ref_heads_reduce_kernel,
which plan_lds (msda_d32.hip) selects from N*Lq*M >= 32768 items.  Every layer of the cfg-4 training step takes them.

Checked against (i) the C oracle fed with the sampling locations / attention weights the prologue forward returns, with the
chain rule of models/ops/modules/ms_deform_attn.py:101-108 applied to the oracle's gradients on the host, (ii) the fp64
composition of the plain function, (iii) themselves under the deterministic flag, with bf16 rows, and without the caller
scratch (the kernels that need none).  All sizes are the bench's own (BASELINE configs[1], configs[3] per rank)."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

C2 = [(48, 48), (24, 24), (12, 12), (6, 6)]
C4 = [(28, 28), (14, 14), (7, 7), (4, 4)]
BIG = {                                   # name: (N, Lq, shapes) — all at M = 8, P = 4: N*Lq*M >= 32768
    "cfg2_encoder": (2, 3060, C2),        # 48 960 items
    "cfg4_decoder": (32, 300, C4),        # 76 800
    "cfg4_encoder": (32, 1045, C4),       # 267 520
}
M, P, D = 8, 4, 32


def _case(name, seed=0):
    N, Lq, shapes = BIG[name]
    g = torch.Generator().manual_seed(1234 + seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    sh = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    value = torch.rand(N, S, M, D, generator=g) - 0.5
    ref = torch.rand(N, Lq, L, 2, generator=g) * 1.2 - 0.1                # some reference points outside the maps
    off = torch.randn(N, Lq, M, L, P, 2, generator=g) * 2.0               # pixels, as the module's projection emits them
    logits = torch.randn(N, Lq, M, L * P, generator=g)
    go = torch.randn(N, Lq, M * D, generator=g)
    return sh, lsi, value, ref, off, logits, go


def _plan(native, name, **kw):
    N, Lq, shapes = BIG[name]
    S = sum(h * w for h, w in shapes)
    return native.describe_plan(N, S, M, D, len(shapes), Lq, P, prologue=True, **kw)


@pytest.fixture(scope="module")
def native():
    from uvhand_amd import _native
    _native.load()
    return _native


@pytest.mark.parametrize("name", list(BIG))
def test_these_sizes_take_the_lds_stage_kernels(native, name):
    """The premise of this file: with caller scratch the plan is the LDS-stage forward + the fused LDS-stage backward with
    the per-head reference-point reduction; without scratch the backward falls back to the tiled fused kernel."""
    plan = _plan(native, name)
    assert "fwd=lds(" in plan and "bwd=fused_lds(" in plan and "heads_reduce" in plan, plan
    plan = _plan(native, name, has_workspace=False)
    assert "fwd=lds(" in plan and "bwd=fused(" in plan, plan


def _host_chain_rule(sh, loc, attn, g_loc, g_attn):
    """Gradients of the RAW tensors from the oracle's gradients of loc / attn (numpy, fp64):
    loc = ref + off / (W, H)  ->  d off = d loc / (W, H),  d ref = sum over heads and points of d loc;
    attn = softmax(logits) over the L*P points of a (query, head)  ->  d logits = attn * (d attn - <attn, d attn>)."""
    wh = np.stack([sh[:, 1], sh[:, 0]], -1).astype(np.float64)[None, None, None, :, None, :]
    g_loc = g_loc.astype(np.float64)
    g_off = g_loc / wh
    g_ref = g_loc.sum(axis=(2, 4))                                        # [N, Lq, L, 2]
    a = attn.astype(np.float64).reshape(attn.shape[0], attn.shape[1], attn.shape[2], -1)
    ga = g_attn.astype(np.float64).reshape(a.shape)
    g_logits = a * (ga - (a * ga).sum(-1, keepdims=True))
    return g_off, g_logits, g_ref


@pytest.mark.parametrize("use_workspace", [True, False], ids=["scratch", "no_scratch"])
@pytest.mark.parametrize("name", list(BIG))
def test_prologue_lds_stage_against_the_c_oracle(native, oracle, name, use_workspace):
    sh, lsi, value, ref, off, logits, go = _case(name)
    dv = lambda t: t.cuda()
    out, loc, attn = native.ms_deform_attn_forward_prologue(dv(value), dv(sh), dv(lsi), dv(ref), dv(off), dv(logits), 64)
    gv, goff, glog, gref = native.ms_deform_attn_backward_prologue(dv(value), dv(sh), dv(lsi), loc, attn, dv(go),
                                                                   use_workspace=use_workspace)
    torch.cuda.synchronize()
    # the prologue itself (modules/ms_deform_attn.py:101-108) against fp64 on the host
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).double()
    loc64 = ref.double()[:, :, None, :, None, :] + off.double() / wh[None, None, None, :, None, :]
    attn64 = torch.softmax(logits.double(), -1).view(attn.shape)
    assert (loc.cpu().double() - loc64).abs().max().item() < 2e-6          # fp32 add + divide on values of magnitude <= ~2
    assert rel_err(attn.cpu().numpy(), attn64.numpy()) < 2e-6
    # the sampling: the oracle on exactly the fp32 loc / attn the kernels used
    shn, lsin = sh.numpy(), lsi.numpy()
    args = [value.numpy(), shn, lsin, loc.cpu().numpy(), attn.cpu().numpy()]
    assert rel_err(out.cpu().numpy(), oracle.forward(*args)) < 5e-6
    r_gv, r_gl, r_ga = oracle.backward(go.numpy(), *args)
    assert rel_err(gv.cpu().numpy(), r_gv) < 2e-5
    r_off, r_log, r_ref = _host_chain_rule(shn, args[3], args[4], r_gl, r_ga)
    assert rel_err(glog.cpu().numpy(), r_log.reshape(glog.shape)) < 2e-5
    # location gradients jump where a pixel coordinate is an integer: leave out points within 1e-5 px of one (and, for
    # the reference points, the (query, level) cells that contain such a point)
    pix = args[3].astype(np.float64) * np.stack([shn[:, 1], shn[:, 0]], -1).astype(np.float64)[None, None, None, :, None, :] - 0.5
    near = (np.abs(pix - np.round(pix)) < 1e-5).any(-1)                    # [N, Lq, M, L, P]
    keep = ~near
    assert keep.mean() > 0.999
    assert rel_err(goff.cpu().numpy()[keep], r_off[keep]) < 2e-5
    keep_cell = ~near.any(axis=(2, 4))                                     # [N, Lq, L]
    assert rel_err(gref.cpu().numpy()[keep_cell], r_ref[keep_cell]) < 2e-5


@pytest.mark.parametrize("name", list(BIG))
def test_prologue_lds_stage_scratch_and_no_scratch_agree_bitwise_on_role_a(native, name):
    """With and without scratch two different role-A kernels run (LDS stage + heads reduce / tiled with the in-workgroup
    head sum): the offset and logit gradients are bit-identical (same per-item arithmetic, msda_d32.hip tap_sums), the
    reference-point gradient adds the same per-head terms in head order in both."""
    sh, lsi, value, ref, off, logits, go = _case(name, seed=1)
    dv = lambda t: t.cuda()
    out, loc, attn = native.ms_deform_attn_forward_prologue(dv(value), dv(sh), dv(lsi), dv(ref), dv(off), dv(logits), 64)
    a = native.ms_deform_attn_backward_prologue(dv(value), dv(sh), dv(lsi), loc, attn, dv(go), use_workspace=True)
    b = native.ms_deform_attn_backward_prologue(dv(value), dv(sh), dv(lsi), loc, attn, dv(go), use_workspace=False)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert rel_err(a[3].cpu().numpy(), b[3].cpu().numpy()) < 1e-6
    assert rel_err(a[0].cpu().numpy(), b[0].cpu().numpy()) < 1e-6          # grad_value: summation order only


@pytest.mark.parametrize("name", list(BIG))
def test_prologue_lds_stage_matches_fp64_composition(name):
    """tests/test_module_gpu.py::test_prologue_function_matches_unfused_composition at the training sizes."""
    from uvhand_amd.functions import MSDeformAttnFunction, MSDeformAttnPrologueFunction
    sh, lsi, value, ref, off, logits, go = _case(name, seed=2)
    N, Lq, shapes = BIG[name]
    L = len(shapes)
    sh, lsi, go = sh.cuda(), lsi.cuda(), go.cuda()
    value, ref, off, logits = (t.cuda().requires_grad_(True) for t in (value, ref, off, logits))
    out = MSDeformAttnPrologueFunction.apply(value, sh, lsi, ref, off, logits, 64)
    out.backward(go)
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).double()
    vd, rd, od, ld = (t.detach().double().requires_grad_(True) for t in (value, ref, off, logits))
    attn = torch.softmax(ld, -1).view(N, Lq, M, L, P)
    loc = rd[:, :, None, :, None, :] + od / wh[None, None, None, :, None, :]
    out_ref = MSDeformAttnFunction.apply(vd, sh, lsi, loc, attn, 64)
    out_ref.backward(go.double())
    assert rel_err(out.detach().cpu().numpy(), out_ref.detach().cpu().numpy()) < 2e-5
    # A location formed in fp32 differs from the fp64 one by up to 1e-5 px; a point that close to a pixel centre sits in
    # another bilinear cell in the two evaluations and its LOCATION gradient jumps (millions of points here: a few dozen
    # do).  Those points — and the (query, level) cells of the reference-point gradient that contain one — are left out;
    # value and logit gradients are continuous there and are compared everywhere.  Gradients: 1e-4 of max.
    pix = loc.detach() * wh[None, None, None, :, None, :] - 0.5
    near = ((pix - pix.round()).abs() < 2e-5).any(-1)                     # [N, Lq, M, L, P]
    assert near.float().mean().item() < 1e-3
    keep_pt = (~near).cpu().numpy()
    keep_cell = (~near.any(dim=4).any(dim=2)).cpu().numpy()               # [N, Lq, L]
    for nm, g32, t64 in zip(("value", "ref", "offsets", "logits"), (value.grad, ref.grad, off.grad, logits.grad), (vd, rd, od, ld)):
        a, b = g32.cpu().numpy(), t64.grad.cpu().numpy()
        if nm == "ref":
            a, b = a[keep_cell], b[keep_cell]
        elif nm == "offsets":
            a, b = a[keep_pt], b[keep_pt]
        e = rel_err(a, b)
        assert e < 1e-4, (nm, e)


@pytest.mark.parametrize("name", list(BIG))
def test_merged_projection_layout_on_the_lds_stage_kernels(name):
    """Offsets and logits read in place from one [N, Lq, 3*M*L*P] projection output, gradients written into one tensor of
    that layout (what the module's default path passes): bit-identical to the dense tensors on these kernels too."""
    from uvhand_amd.functions import MSDeformAttnMergedPrologueFunction, MSDeformAttnPrologueFunction
    sh, lsi, value, ref, off, logits, go = _case(name, seed=3)
    N, Lq, shapes = BIG[name]
    L = len(shapes)
    sh, lsi, go = sh.cuda(), lsi.cuda(), go.cuda()
    value, ref, off, logits = (t.cuda().requires_grad_(True) for t in (value, ref, off, logits))
    out = MSDeformAttnPrologueFunction.apply(value, sh, lsi, ref, off, logits, 64)
    out.backward(go)
    want = [out.detach().clone()] + [t.grad.clone() for t in (value, ref, off, logits)]
    for t in (value, ref, off, logits):
        t.grad = None
    proj = torch.cat([off.detach().reshape(N, Lq, -1), logits.detach().reshape(N, Lq, -1)], -1).requires_grad_(True)
    out2 = MSDeformAttnMergedPrologueFunction.apply(value, sh, lsi, ref, proj, 64, M, L, P)
    out2.backward(go)
    mlp = M * L * P
    got = [out2.detach(), value.grad, ref.grad, proj.grad[..., :2 * mlp].reshape(off.shape), proj.grad[..., 2 * mlp:].reshape(logits.shape)]
    for nm, a, b in zip(("out", "value", "ref", "offsets", "logits"), got, want):
        if nm == "value":
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
        else:
            assert torch.equal(a, b), nm


@pytest.mark.parametrize("name", list(BIG))
def test_deterministic_flag_on_the_lds_stage_prologue_kernels(native, name):
    """MSDA_FLAG_DETERMINISTIC through the fused-prologue entry point at these sizes (bwd_fused_lds_d32_kernel<..., DET>):
    bitwise equal run to run, and equal to the default mode up to grad_value's summation order."""
    sh, lsi, value, ref, off, logits, go = _case(name, seed=4)
    dv = lambda t: t.cuda()
    assert "det" in _plan(native, name, deterministic=True)
    out, loc, attn = native.ms_deform_attn_forward_prologue(dv(value), dv(sh), dv(lsi), dv(ref), dv(off), dv(logits), 64)
    base = native.ms_deform_attn_backward_prologue(dv(value), dv(sh), dv(lsi), loc, attn, dv(go))
    runs = [native.ms_deform_attn_backward_prologue(dv(value), dv(sh), dv(lsi), loc, attn, dv(go), deterministic=True)
            for _ in range(3)]
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)
    assert rel_err(runs[0][0].cpu().numpy(), base[0].cpu().numpy()) < 1e-6
    for a, b in zip(runs[0][1:], base[1:]):
        assert torch.equal(a, b)


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("name", list(BIG))
def test_bf16_rows_on_the_lds_stage_prologue_kernels(native, oracle, name):
    """msda_forward_prologue_bf16 / msda_backward_prologue_bf16_gv32 at these sizes against the C oracle on the bf16-rounded
    value / grad_out (tolerances of tests/test_parity_gpu.py::test_bf16_storage_*: the sampled output is rounded to bf16
    once — 4e-3 of max; grad_value and the raw gradients are fp32 — 2e-5 / 5e-5)."""
    sh, lsi, value, ref, off, logits, go = _case(name, seed=5)
    value, go = _bf16_round(value), _bf16_round(go)
    dv = lambda t: t.cuda()
    out, loc, attn = native.ms_deform_attn_forward_prologue(dv(value).to(torch.bfloat16), dv(sh), dv(lsi), dv(ref), dv(off),
                                                            dv(logits), 64)
    assert out.dtype == torch.bfloat16
    gv, goff, glog, gref = native.ms_deform_attn_backward_prologue(dv(value).to(torch.bfloat16), dv(sh), dv(lsi), loc, attn,
                                                                   dv(go).to(torch.bfloat16))
    assert gv.dtype == torch.float32
    torch.cuda.synchronize()
    shn = sh.numpy()
    args = [value.numpy(), shn, lsi.numpy(), loc.cpu().numpy(), attn.cpu().numpy()]
    assert rel_err(out.float().cpu().numpy(), oracle.forward(*args)) < 4e-3
    r_gv, r_gl, r_ga = oracle.backward(go.numpy(), *args)
    assert rel_err(gv.cpu().numpy(), r_gv) < 2e-5
    r_off, r_log, r_ref = _host_chain_rule(shn, args[3], args[4], r_gl, r_ga)
    assert rel_err(glog.cpu().numpy(), r_log.reshape(glog.shape)) < 5e-5
    pix = args[3].astype(np.float64) * np.stack([shn[:, 1], shn[:, 0]], -1).astype(np.float64)[None, None, None, :, None, :] - 0.5
    near = (np.abs(pix - np.round(pix)) < 1e-5).any(-1)
    assert rel_err(goff.cpu().numpy()[~near], r_off[~near]) < 5e-5
    keep_cell = ~near.any(axis=(2, 4))
    assert rel_err(gref.cpu().numpy()[keep_cell], r_ref[keep_cell]) < 5e-5
