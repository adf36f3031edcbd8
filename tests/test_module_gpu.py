"""Module-level parity on the GPU: uvhand_amd.modules.MSDeformAttn against golden vectors captured from
the reference module (models/ops/modules/ms_deform_attn.py:80-140) with a fixed state_dict."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _module():
    from uvhand_amd.modules import MSDeformAttn
    mod = MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)
    state = load_golden("module_state")
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    return mod.cuda()


@pytest.mark.parametrize("case", ["module_2d", "module_42d"])
def test_module_forward_backward_match_reference(case):
    z = load_golden(case)
    mod = _module()
    query = torch.from_numpy(z["query"]).cuda().requires_grad_(True)
    src = torch.from_numpy(z["src"]).cuda().requires_grad_(True)
    refp = torch.from_numpy(z["refp"]).cuda().requires_grad_(True)
    out = mod(query, refp, src, torch.from_numpy(z["shapes"]).cuda(), torch.from_numpy(z["level_start"]).cuda(),
              torch.from_numpy(z["mask"]).cuda())
    out.backward(torch.from_numpy(z["gout"]).cuda())
    torch.cuda.synchronize()
    # fp32 module (GEMMs on hipBLASLt) vs the reference's fp32 run on CPU: 1e-4 of each tensor's max
    assert rel_err(out.detach().cpu().numpy(), z["out"]) < 1e-4
    assert rel_err(query.grad.cpu().numpy(), z["grad_query"]) < 2e-4
    assert rel_err(src.grad.cpu().numpy(), z["grad_src"]) < 2e-4
    assert rel_err(refp.grad.cpu().numpy(), z["grad_refp"]) < 2e-4
    for name, p in mod.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), z["pgrad." + name]) < 3e-4, name


def test_module_4d_branch_matches_the_dn_dab_reference_module():
    """models/dn_dab_dino_deformable_detr/ops/modules/ms_deform_attn.py:105-108, imported and run by
    tests/golden/gen_golden_r02.py with the shared module_state weights."""
    z = load_golden("module_4d")
    mod = _module()
    query = torch.from_numpy(z["query"]).cuda().requires_grad_(True)
    src = torch.from_numpy(z["src"]).cuda().requires_grad_(True)
    refp = torch.from_numpy(z["refp"]).cuda().requires_grad_(True)
    out = mod(query, refp, src, torch.from_numpy(z["shapes"]).cuda(), torch.from_numpy(z["level_start"]).cuda(),
              torch.from_numpy(z["mask"]).cuda())
    out.backward(torch.from_numpy(z["gout"]).cuda())
    torch.cuda.synchronize()
    assert rel_err(out.detach().cpu().numpy(), z["out"]) < 1e-4
    assert rel_err(query.grad.cpu().numpy(), z["grad_query"]) < 2e-4
    assert rel_err(src.grad.cpu().numpy(), z["grad_src"]) < 2e-4
    assert rel_err(refp.grad.cpu().numpy(), z["grad_refp"]) < 2e-4
    for name, p in mod.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), z["pgrad." + name]) < 3e-4, name


def test_module_4d_reference_boxes_run_and_are_finite():
    """The upstream box branch (dn_dab copy, ops/modules/ms_deform_attn.py:106-108)."""
    mod = _module()
    g = torch.Generator().manual_seed(0)
    shapes = torch.tensor([[8, 8], [4, 4], [2, 2], [1, 1]], dtype=torch.long).cuda()
    lsi = torch.tensor([0, 64, 80, 84], dtype=torch.long).cuda()
    q = torch.randn(2, 10, 256, generator=g).cuda()
    src = torch.randn(2, 85, 256, generator=g).cuda()
    box = torch.rand(2, 10, 4, 4, generator=g).cuda()
    out = mod(q, box, src, shapes, lsi)
    # same numbers as the 2-d branch with explicitly scaled offsets
    off = mod.sampling_offsets(q).view(2, 10, 8, 4, 4, 2)
    loc = box[:, :, None, :, None, :2] + off / 4 * box[:, :, None, :, None, 2:] * 0.5
    from uvhand_amd.functions import MSDeformAttnFunction
    aw = torch.softmax(mod.attention_weights(q).view(2, 10, 8, 16), -1).view(2, 10, 8, 4, 4)
    ref = mod.output_proj(MSDeformAttnFunction.apply(mod.value_proj(src).view(2, 85, 8, 32), shapes, lsi, loc, aw, 64))
    assert torch.isfinite(out).all() and torch.allclose(out, ref, rtol=1e-5, atol=1e-6)


def test_module_runs_under_autocast_like_reference():
    """Under autocast the Linear outputs are half while softmax / reference points stay float: the
    function up-casts value exactly like the reference (functions/ms_deform_attn_func.py:26,37)."""
    mod = _module()
    z = load_golden("module_2d")
    args = [torch.from_numpy(z[k]).cuda() for k in ("query", "refp", "src", "shapes", "level_start")]
    ref = mod(*args)
    with torch.autocast("cuda", dtype=torch.float16):
        out = mod(*args)
    assert torch.isfinite(out).all()
    assert rel_err(out.float().detach().cpu().numpy(), ref.detach().cpu().numpy()) < 2e-2


@pytest.mark.parametrize("case", ["module_2d", "module_42d"])
def test_cpp_module_node_equals_python_composition(case):
    """The fp32 fused path as ONE C++ autograd node (msda_torch.cpp: module_forward, what runs by default when the torch
    extension is built) against the same path composed in Python (cpp_node = False): the same kernels in the same order —
    identical forward, identical gradients except where grad_value's summation order enters (value_proj's parameters and
    the gradient of src: within 1e-5 of max)."""
    from uvhand_amd import _ext
    if _ext.get() is None or not hasattr(_ext.get(), "module_forward"):
        pytest.skip("torch extension not built")
    z = load_golden(case)
    res = []
    for cpp in (True, False):
        mod = _module()
        mod.cpp_node = cpp
        query = torch.from_numpy(z["query"]).cuda().requires_grad_(True)
        src = torch.from_numpy(z["src"]).cuda().requires_grad_(True)
        refp = torch.from_numpy(z["refp"]).cuda().requires_grad_(True)
        out = mod(query, refp, src, torch.from_numpy(z["shapes"]).cuda(), torch.from_numpy(z["level_start"]).cuda(),
                  torch.from_numpy(z["mask"]).cuda())
        assert ("MSDAModuleFunction" in out.grad_fn.name()) == cpp
        out.backward(torch.from_numpy(z["gout"]).cuda())
        res.append((out.detach(), query.grad, refp.grad, src.grad, {n: p.grad for n, p in mod.named_parameters()}))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert rel_err(a[3].cpu().numpy(), b[3].cpu().numpy()) < 1e-5
    for n in a[4]:
        if n.startswith("value_proj"):
            assert rel_err(a[4][n].cpu().numpy(), b[4][n].cpu().numpy()) < 1e-5, n
        else:
            assert torch.equal(a[4][n], b[4][n]), n


@pytest.mark.parametrize("case", ["module_2d", "module_42d"])
def test_cpp_bf16_module_node_equals_python_composition_under_autocast(case):
    """autocast(bfloat16) + bf16 rows as ONE C++ node (module_forward_bf16) against the Python composition of the same steps:
    identical forward and query / reference-point / projection gradients; where grad_value's summation order enters
    (value_proj's parameters, the gradient of src) within bf16 resolution of each other."""
    from uvhand_amd import _ext
    if _ext.get() is None or not hasattr(_ext.get(), "module_forward_bf16"):
        pytest.skip("torch extension not built")
    z = load_golden(case)
    res = []
    for cpp in (True, False):
        mod = _module()
        mod.cpp_node = cpp
        mod.bf16_storage = True
        query = torch.from_numpy(z["query"]).cuda().requires_grad_(True)
        src = torch.from_numpy(z["src"]).cuda().requires_grad_(True)
        refp = torch.from_numpy(z["refp"]).cuda().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = mod(query, refp, src, torch.from_numpy(z["shapes"]).cuda(), torch.from_numpy(z["level_start"]).cuda(),
                      torch.from_numpy(z["mask"]).cuda())
        assert out.dtype == torch.bfloat16
        assert ("MSDAModuleBF16Function" in out.grad_fn.name()) == cpp
        out.backward(torch.from_numpy(z["gout"]).cuda().to(out.dtype))
        res.append((out.detach(), query.grad, refp.grad, src.grad, {n: p.grad for n, p in mod.named_parameters()}))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3].dtype == b[3].dtype == torch.float32 and rel_err(a[3].cpu().numpy(), b[3].cpu().numpy()) < 2e-2
    for n in a[4]:
        assert a[4][n].dtype == torch.float32
        if n.startswith("value_proj"):
            assert rel_err(a[4][n].cpu().numpy(), b[4][n].cpu().numpy()) < 2e-2, n
        else:
            assert torch.equal(a[4][n], b[4][n]), n
    # and against the float32 fixture, at bf16 tolerance
    assert rel_err(a[0].float().cpu().numpy(), z["out"]) < 3e-2


def test_cpp_module_node_under_no_grad_is_the_same_forward():
    """Inference: the one-node form runs its forward only and returns exactly what the training-mode call returns."""
    from uvhand_amd import _ext
    if _ext.get() is None or not hasattr(_ext.get(), "module_forward"):
        pytest.skip("torch extension not built")
    z = load_golden("module_2d")
    mod = _module()
    args = [torch.from_numpy(z[k]).cuda() for k in ("query", "refp", "src", "shapes", "level_start", "mask")]
    ref = mod(*args)
    assert "MSDAModuleFunction" in ref.grad_fn.name()
    with torch.no_grad():
        out = mod(*args)
    assert out.grad_fn is None and not out.requires_grad and torch.equal(out, ref.detach())
    with torch.inference_mode():
        out2 = mod(*args)
    assert torch.equal(out2, ref.detach())
    assert rel_err(out.cpu().numpy(), z["out"]) < 1e-4


def test_half_module_follows_the_dino_amp_branch():
    """module.half() with half inputs: the op runs in float32 and its output returns to half before output_proj
    (models/dino/ops/modules/ms_deform_attn.py:124-131)."""
    mod = _module()
    z = load_golden("module_2d")
    query, refp, src, shapes, lsi = [torch.from_numpy(z[k]).cuda() for k in ("query", "refp", "src", "shapes", "level_start")]
    ref = mod(query, refp, src, shapes, lsi)
    import copy
    hmod = copy.deepcopy(mod).half()
    out = hmod(query.half(), refp.half(), src.half(), shapes, lsi)
    assert out.dtype == torch.float16 and torch.isfinite(out).all()
    assert rel_err(out.float().detach().cpu().numpy(), ref.detach().cpu().numpy()) < 3e-2
    out.float().sum().backward()
    assert all(p.grad is not None and p.grad.dtype == torch.float16 for p in hmod.parameters())


@pytest.mark.parametrize("autocast", [False, True])
def test_module_bf16_storage_takes_the_fused_path_and_tracks_fp32(autocast, monkeypatch):
    """bf16_storage keeps the fast module path (fused prologue on bf16 rows + merged projection, msda_*_prologue_bf16*):
    forward and EVERY gradient within bf16 tolerance of the fp32 module (2e-2 of each tensor's max: one rounding of value,
    of the sampled output and of grad_out to 8 significant bits; locations, weights and accumulation stay fp32)."""
    from uvhand_amd import _native
    calls = {"fwd": 0, "bwd": 0}
    for name, key in (("ms_deform_attn_forward_prologue", "fwd"), ("ms_deform_attn_backward_prologue", "bwd")):
        orig = getattr(_native, name)

        def wrapped(*a, _o=orig, _k=key, **kw):
            assert a[0].dtype == torch.bfloat16                      # rows really are bf16
            calls[_k] += 1
            return _o(*a, **kw)
        monkeypatch.setattr(_native, name, wrapped)
    z = load_golden("module_2d")
    tensors = lambda: [torch.from_numpy(z[k]).cuda() for k in ("query", "refp", "src", "shapes", "level_start", "mask")]

    def run(bf16):
        mod = _module()
        mod.bf16_storage = bf16
        a = tensors()
        a[0].requires_grad_(True); a[2].requires_grad_(True)
        if bf16 and autocast:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = mod(*a)
        else:
            out = mod(*a)
        out.float().backward(torch.from_numpy(z["gout"]).cuda())
        return out.float().detach(), a[0].grad, a[2].grad, {n: p.grad for n, p in mod.named_parameters()}

    monkeypatch.undo()
    ref = run(False)
    for name, key in (("ms_deform_attn_forward_prologue", "fwd"), ("ms_deform_attn_backward_prologue", "bwd")):
        orig = getattr(_native, name)

        def wrapped(*a, _o=orig, _k=key, **kw):
            assert a[0].dtype == torch.bfloat16
            calls[_k] += 1
            return _o(*a, **kw)
        monkeypatch.setattr(_native, name, wrapped)
    got = run(True)
    from uvhand_amd import _ext
    one_node = autocast and _ext.get() is not None and hasattr(_ext.get(), "module_forward_bf16")
    # under autocast the whole path is one C++ node (module_forward_bf16, compared with this composition in
    # test_cpp_bf16_module_node_equals_python_composition_under_autocast): the Python entry points are not visited
    assert calls == ({"fwd": 0, "bwd": 0} if one_node else {"fwd": 1, "bwd": 1})
    tol = 4e-2 if autocast else 2e-2                                 # autocast also runs the value / output GEMMs in bf16
    assert rel_err(got[0].cpu().numpy(), ref[0].cpu().numpy()) < tol
    assert rel_err(got[1].cpu().numpy(), ref[1].cpu().numpy()) < tol
    assert rel_err(got[2].cpu().numpy(), ref[2].cpu().numpy()) < tol
    for n in ref[3]:
        assert got[3][n].dtype == torch.float32
        assert rel_err(got[3][n].cpu().numpy(), ref[3][n].cpu().numpy()) < tol, n


# ---------------------------------------------------------------------------------------------
# fused prologue (softmax + location arithmetic inside the kernels) vs the unfused composition
# ---------------------------------------------------------------------------------------------
def _prologue_case(seed, N, Lq, shapes, M=8, P=4):
    g = torch.Generator().manual_seed(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    sh = torch.tensor(shapes, dtype=torch.long).cuda()
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    value = (torch.rand(N, S, M, 32, generator=g) - 0.5).cuda().requires_grad_(True)
    ref = (torch.rand(N, Lq, L, 2, generator=g) * 1.4 - 0.2).cuda().requires_grad_(True)
    off = (torch.randn(N, Lq, M, L, P, 2, generator=g) * 2.0).cuda().requires_grad_(True)
    logits = torch.randn(N, Lq, M, L * P, generator=g).cuda().requires_grad_(True)
    go = torch.randn(N, Lq, M * 32, generator=g).cuda()
    return sh, lsi, value, ref, off, logits, go


@pytest.mark.parametrize("N,Lq,shapes", [(2, 37, [(12, 12), (6, 6), (3, 3), (2, 2)]),
                                          (2, 300, [(48, 48), (24, 24), (12, 12), (6, 6)]),
                                          (1, 2100, [(16, 16), (8, 8), (4, 4), (2, 2)]),       # multi-pass role B
                                          (3, 5, [(5, 7), (3, 4)])])
def test_prologue_function_matches_unfused_composition(N, Lq, shapes):
    from uvhand_amd import _native
    from uvhand_amd.functions import MSDeformAttnFunction, MSDeformAttnPrologueFunction
    sh, lsi, value, ref, off, logits, go = _prologue_case(N + Lq, N, Lq, shapes)
    L, P = len(shapes), 4
    assert _native.prologue_supported(value, ref, off, logits)
    out = MSDeformAttnPrologueFunction.apply(value, sh, lsi, ref, off, logits, 64)
    out.backward(go)
    got = [t.grad.clone() for t in (value, ref, off, logits)]
    for t in (value, ref, off, logits):
        t.grad = None
    # the reference's own arithmetic (modules/ms_deform_attn.py:101-108) in fp64 on the plain function
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).double()
    vd, rd, od, ld = (t.detach().double().requires_grad_(True) for t in (value, ref, off, logits))
    attn = torch.softmax(ld, -1).view(N, Lq, 8, L, P)
    loc = rd[:, :, None, :, None, :] + od / wh[None, None, None, :, None, :]
    out_ref = MSDeformAttnFunction.apply(vd, sh, lsi, loc, attn, 64)
    out_ref.backward(go.double())
    # fp32 kernels vs the same arithmetic in fp64: forward 2e-5 of max (locations are formed in fp32, so a
    # sample moves by up to 1e-5 px), gradients 1e-4 of max
    e_out = rel_err(out.detach().cpu().numpy(), out_ref.detach().cpu().numpy())
    assert e_out < 2e-5, e_out
    for name, g32, t64 in zip(("value", "ref", "offsets", "logits"), got, (vd, rd, od, ld)):
        e = rel_err(g32.cpu().numpy(), t64.grad.cpu().numpy())
        assert e < 1e-4, (name, e)


def test_module_fused_and_unfused_prologue_agree():
    z = load_golden("module_42d")
    mod = _module()
    args = lambda: [torch.from_numpy(z[k]).cuda() for k in ("query", "refp", "src", "shapes", "level_start", "mask")]
    res = {}
    for fused in (True, False):
        mod.fused_prologue = fused
        mod.zero_grad()
        a = args()
        a[0].requires_grad_(True); a[1].requires_grad_(True); a[2].requires_grad_(True)
        out = mod(*a)
        out.backward(torch.from_numpy(z["gout"]).cuda())
        res[fused] = [out.detach()] + [t.grad for t in a[:3]] + [p.grad.clone() for p in mod.parameters()]
    for x, y in zip(res[True], res[False]):
        assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("N,Lq,shapes", [(2, 300, [(48, 48), (24, 24), (12, 12), (6, 6)]),
                                          (3, 5, [(5, 7), (3, 4)])])
def test_merged_projection_function_equals_separate_tensors(N, Lq, shapes):
    """Offsets and logits read in place from one [N, Lq, 3*M*L*P] projection output (row stride != dense
    width), gradients written back into one tensor of that layout: bit-identical to the dense-tensor path
    (grad_value up to summation order — the order of a row's records after the counting sort's LDS atomics
    is not fixed run to run, as with the reference's atomicAdd scatter)."""
    from uvhand_amd.functions import MSDeformAttnMergedPrologueFunction, MSDeformAttnPrologueFunction
    sh, lsi, value, ref, off, logits, go = _prologue_case(7 * N + Lq, N, Lq, shapes)
    M, L, P = 8, len(shapes), 4
    out = MSDeformAttnPrologueFunction.apply(value, sh, lsi, ref, off, logits, 64)
    out.backward(go)
    want = [out.detach().clone()] + [t.grad.clone() for t in (value, ref, off, logits)]
    for t in (value, ref, off, logits):
        t.grad = None
    proj = torch.cat([off.detach().reshape(N, Lq, -1), logits.detach().reshape(N, Lq, -1)], -1).requires_grad_(True)
    out2 = MSDeformAttnMergedPrologueFunction.apply(value, sh, lsi, ref, proj, 64, M, L, P)
    out2.backward(go)
    mlp = M * L * P
    got = [out2.detach(), value.grad, ref.grad, proj.grad[..., :2 * mlp].reshape(off.shape),
           proj.grad[..., 2 * mlp:].reshape(logits.shape)]
    for name, a, b in zip(("out", "value", "ref", "offsets", "logits"), got, want):
        if name == "value":
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
        else:
            assert torch.equal(a, b), name


def test_prologue_rejects_bad_row_layouts():
    from uvhand_amd import _native
    sh, lsi, value, ref, off, logits, _ = _prologue_case(5, 2, 6, [(5, 7), (3, 4)])
    with pytest.raises(RuntimeError, match="contiguous within a query row"):
        _native.ms_deform_attn_forward_prologue(value, sh, lsi, ref, off.detach().transpose(3, 4).contiguous().transpose(3, 4),
                                                logits.detach(), 64)
    wide = torch.zeros(2, 6, 8 * 2 * 4 * 2 + 1, device="cuda")           # odd row stride for the (x, y) pairs
    with pytest.raises(RuntimeError, match="row strides"):
        _native.ms_deform_attn_forward_prologue(value, sh, lsi, ref, wide[..., :128].view(2, 6, 8, 2, 4, 2),
                                                logits.detach(), 64)


def test_module_merged_and_separate_projections_agree():
    z = load_golden("module_2d")
    mod = _module()
    args = lambda: [torch.from_numpy(z[k]).cuda() for k in ("query", "refp", "src", "shapes", "level_start", "mask")]
    res = {}
    for merged in (True, False):
        mod.merged_projection = merged
        mod.zero_grad()
        a = args()
        a[0].requires_grad_(True); a[1].requires_grad_(True); a[2].requires_grad_(True)
        out = mod(*a)
        out.backward(torch.from_numpy(z["gout"]).cuda())
        res[merged] = [out.detach()] + [t.grad for t in a[:3]] + [p.grad.clone() for p in mod.parameters()]
    for x, y in zip(res[True], res[False]):
        assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < 1e-5


def test_caller_side_argument_construction_feeds_the_module():
    """A feature pyramid with padded samples -> flatten_feature_levels / encoder_reference_points (uvhand_amd.utils,
    the reference transformer's glue) -> MSDeformAttn on the GPU: shapes, finiteness, and zero influence of the
    padded pixels' features (they are masked out of value)."""
    from uvhand_amd.modules import MSDeformAttn
    from uvhand_amd.utils import encoder_reference_points, flatten_feature_levels
    torch.manual_seed(0)
    shapes, N, C = [(12, 16), (6, 8), (3, 4), (2, 2)], 2, 256
    srcs = [torch.randn(N, C, h, w, device="cuda") for h, w in shapes]
    poss = [torch.randn(N, C, h, w, device="cuda") * 0.1 for h, w in shapes]
    masks = []
    for h, w in shapes:
        m = torch.zeros(N, h, w, dtype=torch.bool, device="cuda")
        m[1, :, w - w // 4:] = True                                       # sample 1: right quarter is padding
        masks.append(m)
    level_embed = torch.randn(len(shapes), C, device="cuda") * 0.1
    mod = MSDeformAttn(C, len(shapes), 8, 4).cuda()
    src, mask, pos, ss, lsi, valid = flatten_feature_levels(srcs, masks, poss, level_embed)
    ref = encoder_reference_points(ss, valid)
    assert ss.is_cuda and ss.dtype == torch.int64 and ref.shape == (N, src.shape[1], len(shapes), 2)
    out = mod(src + pos, ref, src, ss, lsi, mask)
    assert out.shape == src.shape and torch.isfinite(out).all()
    src2 = src.clone()
    src2[mask] = 1e6                                                       # garbage in the padded pixels' features
    out2 = mod(src + pos, ref, src2, ss, lsi, mask)
    assert torch.equal(out, out2)


def _random_prologue_geometries(count, seed):
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < count:
        L, P, M = int(rng.choice([1, 2, 4])), int(rng.choice([1, 2, 4])), int(rng.choice([1, 2, 4, 8]))
        shapes = [(int(rng.randint(1, 20)), int(rng.randint(1, 20))) for _ in range(L)]
        out.append((int(rng.randint(1, 4)), int(rng.choice([1, 5, 64, 300, 500])), shapes, M, P))
    return out


@pytest.mark.parametrize("idx,geo", list(enumerate(_random_prologue_geometries(16, 7))))
def test_prologue_random_geometries(idx, geo):
    """Fused prologue (merged projection layout) on random supported geometries against the fp64 composition of
    the plain function; geometries the kernels do not take must be reported unsupported, not mis-computed."""
    from uvhand_amd import _native
    from uvhand_amd.functions import MSDeformAttnFunction, MSDeformAttnMergedPrologueFunction
    N, Lq, shapes, M, P = geo
    sh, lsi, value, ref, off, logits, go = _prologue_case(900 + idx, N, Lq, shapes, M=M, P=P)
    L = len(shapes)
    if not _native.prologue_supported(value, ref, off, logits):
        pytest.skip("geometry not taken by the fused-prologue kernels (the module composes the plain ops)")
    proj = torch.cat([off.detach().reshape(N, Lq, -1), logits.detach().reshape(N, Lq, -1)], -1).requires_grad_(True)
    out = MSDeformAttnMergedPrologueFunction.apply(value, sh, lsi, ref, proj, 64, M, L, P)
    out.backward(go)
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).double()
    vd, rd, pd = (t.detach().double().requires_grad_(True) for t in (value, ref, proj))
    mlp = M * L * P
    od, ld = pd[..., :2 * mlp].reshape(N, Lq, M, L, P, 2), pd[..., 2 * mlp:].reshape(N, Lq, M, L * P)
    attn = torch.softmax(ld, -1).view(N, Lq, M, L, P)
    loc = rd[:, :, None, :, None, :] + od / wh[None, None, None, :, None, :]
    out_ref = MSDeformAttnFunction.apply(vd, sh, lsi, loc, attn, 64)
    out_ref.backward(go.double())
    assert rel_err(out.detach().cpu().numpy(), out_ref.detach().cpu().numpy()) < 2e-5
    for name, g32, t64 in zip(("value", "ref", "projected"), (value.grad, ref.grad, proj.grad), (vd, rd, pd)):
        assert rel_err(g32.cpu().numpy(), t64.grad.cpu().numpy()) < 1e-4, (name, geo)


@pytest.mark.parametrize("seed", range(10))
def test_module_random_configurations_fused_vs_composed(seed):
    """MSDeformAttn with random (heads, levels, points, queries, reference-point width, padding mask): the default
    path (fused prologue, merged projection, masked-row kernels, custom weight gradient) against the same module
    composing everything in PyTorch around the plain op (fused_prologue = False)."""
    from uvhand_amd.modules import MSDeformAttn
    rng = np.random.RandomState(1000 + seed)
    M, L, P = int(rng.choice([1, 2, 4, 8])), int(rng.choice([1, 2, 3, 4])), int(rng.choice([1, 2, 4]))
    shapes = [(int(rng.randint(2, 16)), int(rng.randint(2, 16))) for _ in range(L)]
    N, Lq, width = int(rng.randint(1, 4)), int(rng.choice([1, 3, 50, 300])), int(rng.choice([2, 42]))
    S, C = sum(h * w for h, w in shapes), 32 * M
    torch.manual_seed(seed)
    mod = MSDeformAttn(C, L, M, P).cuda()
    with torch.no_grad():
        for p in mod.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    sh = torch.tensor(shapes, dtype=torch.long).cuda()
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    query = torch.randn(N, Lq, C).cuda()
    src = torch.randn(N, S, C).cuda()
    refp = torch.rand(N, Lq, L, width).cuda()
    mask = (torch.rand(N, S) < 0.1).cuda() if seed % 2 else None
    go = torch.randn(N, Lq, C).cuda()
    res = {}
    for fused in (True, False):
        mod.fused_prologue = fused
        mod.zero_grad()
        q, s_, r = (t.clone().requires_grad_(True) for t in (query, src, refp))
        out = mod(q, r, s_, sh, lsi, mask)
        out.backward(go)
        res[fused] = [out.detach(), q.grad, s_.grad, r.grad] + [p.grad.clone() for p in mod.parameters()]
    for x, y in zip(res[True], res[False]):
        assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < 5e-5, (M, L, P, shapes, N, Lq, width)


def test_whole_module_forward_backward_in_a_hip_graph():
    """One MSDeformAttn forward + backward (projections, fused prologue, kernels, weight gradients) captured into a HIP
    graph and replayed on new data: nothing in the module or the library synchronises, allocates outside the capture
    pool or reads spatial_shapes back (the shapes-sum assert is skipped during capture — modules/ms_deform_attn.py)."""
    z = load_golden("module_2d")
    mod = _module()
    shapes, lsi = torch.from_numpy(z["shapes"]).cuda(), torch.from_numpy(z["level_start"]).cuda()
    mask = torch.from_numpy(z["mask"]).cuda()
    query = torch.zeros_like(torch.from_numpy(z["query"])).cuda().requires_grad_(True)
    src = torch.zeros_like(torch.from_numpy(z["src"])).cuda().requires_grad_(True)
    refp = torch.zeros_like(torch.from_numpy(z["refp"])).cuda()
    gout = torch.zeros_like(torch.from_numpy(z["gout"])).cuda()

    def step():
        for t in (query, src, *mod.parameters()):
            t.grad = None
        out = mod(query, refp, src, shapes, lsi, mask)
        out.backward(gout)
        return out

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for _ in range(3):                      # warm-up on the side stream (allocator, hipBLASLt workspaces)
            step()
        stream.synchronize()
        fresh_shapes = shapes.clone()           # a shapes tensor the module has never seen: no read-back during capture
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for t in (query, src, *mod.parameters()):
                t.grad = None
            out = mod(query, refp, src, fresh_shapes, lsi, mask)
            out.backward(gout)
        with torch.no_grad():                   # new data into the captured buffers
            query.copy_(torch.from_numpy(z["query"])); src.copy_(torch.from_numpy(z["src"]))
            refp.copy_(torch.from_numpy(z["refp"])); gout.copy_(torch.from_numpy(z["gout"]))
        graph.replay()
        graph.replay()
    torch.cuda.synchronize()
    assert rel_err(out.detach().cpu().numpy(), z["out"]) < 1e-4
    assert rel_err(query.grad.cpu().numpy(), z["grad_query"]) < 2e-4
    assert rel_err(src.grad.cpu().numpy(), z["grad_src"]) < 2e-4
    for name, p in mod.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), z["pgrad." + name]) < 3e-4, name


def test_module_under_inference_mode():
    """The reference module has no restriction under torch.inference_mode(); the shapes check must not need a version
    counter (inference tensors have none)."""
    z = load_golden("module_2d")
    mod = _module().eval()
    with torch.inference_mode():
        args = [torch.from_numpy(z[k]).cuda() for k in ("query", "refp", "src", "shapes", "level_start", "mask")]
        out = mod(*args)
        out2 = mod(*args)
    assert rel_err(out.cpu().numpy(), z["out"]) < 1e-4 and torch.equal(out, out2)


@pytest.mark.parametrize("cpp", [True, False], ids=["cpp_node", "python_composition"])
def test_module_at_encoder_geometry_above_the_lds_stage_threshold(cpp):
    """The reference module at ENCODER geometry with N*Lq*M = 33 440 >= 32 768 items (tests/golden/gen_golden_r04.py:
    d_model 256 / 8 heads, N = 4, 28/14/7/4, Lq = S = 1045, padding mask): here the default path runs the fused-prologue
    LDS-stage kernels and ref_heads_reduce_kernel (msda_d32.hip: plan_lds) — what every encoder layer of the cfg-4 training
    step runs.  Inputs are rebuilt from numpy's frozen generator (tests/golden/big_inputs.py), the fixture holds every 4th
    row of the big outputs, per-row sums of all rows, and every parameter gradient."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import big_inputs
    from uvhand_amd import _ext, _native
    z, gold = big_inputs.module_enc_big_inputs(), load_golden("module_enc_big")
    assert np.array_equal(big_inputs.checksums(z), gold["input_checksums"])       # the inputs the reference saw
    if cpp and (_ext.get() is None or not hasattr(_ext.get(), "module_forward")):
        pytest.skip("torch extension not built")
    N, S, C = z["query"].shape
    plan = _native.describe_plan(N, S, 8, 32, 4, S, 4, prologue=True)
    assert "fwd=lds(" in plan and "bwd=fused_lds(" in plan, plan
    mod = _module()
    mod.cpp_node = cpp
    query = torch.from_numpy(z["query"]).cuda().requires_grad_(True)
    src = torch.from_numpy(z["src"]).cuda().requires_grad_(True)
    refp = torch.from_numpy(z["refp"]).cuda().requires_grad_(True)
    out = mod(query, refp, src, torch.from_numpy(z["shapes"]).cuda(), torch.from_numpy(z["level_start"]).cuda(),
              torch.from_numpy(z["mask"]).cuda())
    assert ("MSDAModuleFunction" in out.grad_fn.name()) == cpp
    out.backward(torch.from_numpy(z["gout"]).cuda())
    torch.cuda.synchronize()
    step = big_inputs.ROW_STEP
    for name, t, tol in (("out", out.detach(), 1e-4), ("grad_query", query.grad, 2e-4), ("grad_src", src.grad, 2e-4)):
        assert rel_err(t[:, ::step].cpu().numpy(), gold[name + "_rows"]) < tol, name
        # every row through its sum (256 terms: the tolerance is relative to the largest row sum)
        assert rel_err(t.double().sum(-1).cpu().numpy(), gold[name + "_rowsum"]) < 2 * tol, name
    assert rel_err(refp.grad.cpu().numpy(), gold["grad_refp"]) < 2e-4
    for name, p in mod.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), gold["pgrad." + name]) < 3e-4, name


def _node_inputs(z):
    return (torch.from_numpy(z["query"]).cuda().requires_grad_(True), torch.from_numpy(z["refp"]).cuda().requires_grad_(True),
            torch.from_numpy(z["src"]).cuda().requires_grad_(True), torch.from_numpy(z["shapes"]).cuda(),
            torch.from_numpy(z["level_start"]).cuda(), torch.from_numpy(z["mask"]).cuda())


def test_frozen_projections_cost_no_weight_gradient_launch():
    """The one-node path skips the weight / bias gradient of every projection whose parameters do not require grad (the lr groups
    and --not_use_params of util/settings.py:447-515 freeze layers): fewer launches through the library (msda_launch_count),
    no .grad on the frozen parameters, everything else unchanged."""
    from uvhand_amd import _ext, _native
    if _ext.get() is None or not hasattr(_ext.get(), "module_forward"):
        pytest.skip("torch extension not built")
    z = load_golden("module_2d")
    results, launches = {}, {}
    for frozen in (False, True):
        mod = _module()
        if frozen:
            for p in list(mod.value_proj.parameters()) + list(mod.output_proj.parameters()):
                p.requires_grad_(False)
        q, r, s_, sh, lsi, mask = _node_inputs(z)
        for _ in range(2):                                   # second round: the cached merged projection is in use as well
            mod.zero_grad(set_to_none=True)
            q.grad = r.grad = s_.grad = None
            before = _native.launch_count()
            out = mod(q, r, s_, sh, lsi, mask)
            assert "MSDAModuleFunction" in out.grad_fn.name()
            out.backward(torch.from_numpy(z["gout"]).cuda())
            torch.cuda.synchronize()
            launches[frozen] = _native.launch_count() - before
        results[frozen] = (out.detach(), q.grad, s_.grad, r.grad, {n: p.grad for n, p in mod.named_parameters()})
    assert launches[True] < launches[False], launches
    for a, b in zip(results[True][:4], results[False][:4]):
        assert torch.equal(a, b)
    for n, g in results[True][4].items():
        if n.startswith(("value_proj", "output_proj")):
            assert g is None, n
        else:
            assert torch.equal(g, results[False][4][n]), n


def test_merged_projection_storage_follows_parameter_updates():
    """[sampling_offsets ; attention_weights] is the parameters' own storage (views of one buffer): an optimizer-style in-place
    update, a write through `.data` (invisible to version counters: ADVICE r04), load_state_dict — also with assign=True, which
    re-seats the parameters — and _reset_parameters all show in the next forward, and no concatenation kernel runs."""
    z = load_golden("module_2d")
    mod = _module()
    q, r, s_, sh, lsi, mask = _node_inputs(z)
    out0 = mod(q, r, s_, sh, lsi, mask).detach().clone()
    wm, bm = mod._merged_projection_weights()
    assert wm is not None and mod.attention_weights.bias.data_ptr() == bm.data_ptr() + 4 * mod.sampling_offsets.bias.numel()
    delta = torch.linspace(-1, 1, mod.attention_weights.bias.numel(), device="cuda")
    mod.attention_weights.bias.data.copy_(mod.attention_weights.bias.data + delta)     # `.data` write: no version bump
    out1 = mod(q, r, s_, sh, lsi, mask).detach().clone()
    assert not torch.equal(out0, out1)
    with torch.no_grad():
        mod.attention_weights.bias.sub_(delta)
        mod.attention_weights.bias.add_(delta)
    assert torch.equal(mod(q, r, s_, sh, lsi, mask).detach(), out1)
    ref = _module()                                          # a fresh module with the same update and no cache history
    with torch.no_grad():
        ref.attention_weights.bias.add_(torch.linspace(-1, 1, ref.attention_weights.bias.numel(), device="cuda"))
    assert torch.equal(out1, ref(q, r, s_, sh, lsi, mask).detach())
    mod.load_state_dict(_module().state_dict())
    assert torch.equal(mod(q, r, s_, sh, lsi, mask).detach(), out0)
    assert mod._merged_projection_weights()[0].data_ptr() == wm.data_ptr()                  # copied INTO the views
    mod.load_state_dict(ref.state_dict(), assign=True)                                       # new Parameters: shared again on use
    assert torch.equal(mod(q, r, s_, sh, lsi, mask).detach(), out1)
    assert mod.sampling_offsets.weight.data_ptr() == mod._merged_projection_weights()[0].data_ptr()
    mod.load_state_dict(_module().state_dict())
    torch.manual_seed(5)                                     # (xavier_uniform_ draws the value / output projections)
    mod._reset_parameters()                                  # (re-creates the offsets' bias on the CPU, as the reference's does)
    mod.cuda()
    fresh = _module()
    torch.manual_seed(5)
    fresh._reset_parameters()
    fresh.cuda()
    assert torch.equal(mod(q, r, s_, sh, lsi, mask).detach(), fresh(q, r, s_, sh, lsi, mask).detach())


def test_cpp_node_checks_its_arguments_and_falls_back_in_python():
    """the one-node path validated dtype / device only.  Reference points that merely broadcast ([N, Lq, 1, 2]) and
    a spatial_shapes with fewer rows than n_levels now take the Python composition (which broadcasts like the reference, or
    raises); handed to the node directly they raise instead of reading out of bounds; a create_graph=True backward raises
    like the reference's @once_differentiable."""
    from uvhand_amd import _ext
    ext = _ext.get()
    if ext is None or not hasattr(ext, "module_forward"):
        pytest.skip("torch extension not built")
    z = load_golden("module_2d")
    mod = _module()
    q, r, s_, sh, lsi, mask = _node_inputs(z)
    full = mod(q, r, s_, sh, lsi, mask)
    assert "MSDAModuleFunction" in full.grad_fn.name()
    # broadcastable reference points: same result through the composition
    r1 = r.detach()[:, :, :1].clone()
    out = mod(q, r1.expand(-1, -1, 4, -1), s_, sh, lsi, mask)              # an expanded view is a full-size tensor: the node takes it
    out_b = mod(q, r1, s_, sh, lsi, mask)                                  # [N, Lq, 1, 2]: the module expands it like the reference's broadcast
    assert torch.equal(out_b, out)
    mod.fused_prologue = False                                             # ... and so does the reference's own arithmetic
    out_c = mod(q, r1, s_, sh, lsi, mask)
    mod.fused_prologue = True
    assert rel_err(out_c.detach().cpu().numpy(), out.detach().cpu().numpy()) < 2e-5
    args = (q, r1, s_, mask, sh, lsi, mod.sampling_offsets.weight, mod.sampling_offsets.bias, mod.attention_weights.weight,
            mod.attention_weights.bias, mod.value_proj.weight, mod.value_proj.bias, mod.output_proj.weight, mod.output_proj.bias,
            8, 4, 4, 64, False)
    with pytest.raises(RuntimeError, match="reference points must be"):
        ext.module_forward(*args)
    bad = list(args); bad[1] = r.detach(); bad[4] = sh[:3].contiguous()
    with pytest.raises(RuntimeError, match="spatial_shapes must be"):
        ext.module_forward(*bad)
    # @once_differentiable
    g, = torch.autograd.grad(full.sum(), q, create_graph=True)
    (full * 2).sum().backward(retain_graph=True)                           # a plain second backward of the same graph is fine
    go = torch.ones_like(full).requires_grad_(True)
    with pytest.raises(RuntimeError, match="once_differentiable"):
        torch.autograd.grad(full, q, grad_outputs=go, create_graph=True)


def test_cpp_node_with_two_projection_outputs_falls_back_to_torch_weight_gradients():
    """n_heads * n_levels * n_points == 2 gives the merged projection 6 output rows (2 offsets x 2 + 2 logits), not a
    multiple of 4: the node's weight gradient then takes the torch composition instead of msda_linear_wgrad (which refuses)."""
    from uvhand_amd.modules import MSDeformAttn
    torch.manual_seed(3)
    mod = MSDeformAttn(32, 1, 1, 2).cuda()
    sh = torch.tensor([[6, 5]], dtype=torch.long).cuda()
    lsi = torch.zeros(1, dtype=torch.long).cuda()
    q = torch.randn(2, 7, 32).cuda().requires_grad_(True)
    src = torch.randn(2, 30, 32).cuda().requires_grad_(True)
    ref = torch.rand(2, 7, 1, 2).cuda()
    res = {}
    for cpp in (True, False):
        mod.cpp_node = cpp
        mod.zero_grad(set_to_none=True)
        q.grad = src.grad = None
        out = mod(q, ref, src, sh, lsi)
        out.sum().backward()
        res[cpp] = [out.detach(), q.grad, src.grad] + [p.grad for p in mod.parameters()]
    for a, b in zip(res[True], res[False]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
