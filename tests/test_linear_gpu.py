"""Weight / bias gradient kernel of the bracketing nn.Linear layers (msda_linear_wgrad_f32) against
PyTorch's own fp64 result, and the autograd wrapper against nn.Linear."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(600, 256, 256), (600, 128, 256), (6120, 256, 256), (33440, 128, 256),
                                    (37, 12, 20), (1, 4, 4), (70, 68, 132), (2400, 384, 256), (64, 64, 64)])
def test_wgrad_matches_fp64(M, N, K):
    from uvhand_amd import _native
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).cuda()
    x = torch.randn(M, K, generator=g).cuda()
    gw, gb = _native.linear_wgrad(dy, x)
    ref_w = (dy.double().t() @ x.double()).cpu().numpy()
    ref_b = dy.double().sum(0).cpu().numpy()
    # fp32 MFMA = fp32 fma chain over M terms: error ~ sqrt(M) * 2^-24 of the sum of magnitudes
    assert rel_err(gw.cpu().numpy(), ref_w) < 2e-6
    assert rel_err(gb.cpu().numpy(), ref_b) < 2e-6
    gw2, gb2 = _native.linear_wgrad(dy, x)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)            # fixed-order reduction: reproducible
    gw3, none = _native.linear_wgrad(dy, x, want_bias=False)
    assert none is None and torch.equal(gw, gw3)


def test_wgrad_rejects_what_it_cannot_do():
    from uvhand_amd import _native
    with pytest.raises(RuntimeError):
        _native.linear_wgrad(torch.randn(8, 6).cuda(), torch.randn(8, 8).cuda())     # N % 4 != 0
    with pytest.raises(RuntimeError):
        _native.linear_wgrad(torch.randn(8, 8).cuda().double(), torch.randn(8, 8).cuda().double())
    gw, gb = _native.linear_wgrad(torch.empty(0, 8).cuda(), torch.empty(0, 12).cuda())
    assert not gw.any() and not gb.any() and tuple(gw.shape) == (8, 12)


def test_bracket_linear_equals_nn_linear():
    from uvhand_amd.functions.linear_func import bracket_linear
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 128).cuda()
    x = torch.randn(2, 300, 256).cuda().requires_grad_(True)
    go = torch.randn(2, 300, 128).cuda()
    y = bracket_linear(x, lin)
    y.backward(go)
    got = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None; lin.zero_grad()
    y_ref = lin(x)
    y_ref.backward(go)
    # a small layer: the library's own forward / input-gradient kernels (exact fp32 MFMA) — both sides against fp64
    y64 = x.detach().double() @ lin.weight.detach().double().t() + lin.bias.detach().double()
    gx64 = go.double() @ lin.weight.detach().double()
    assert rel_err(y.detach().cpu().numpy(), y64.cpu().numpy()) < 2e-6 and rel_err(y_ref.detach().cpu().numpy(), y64.cpu().numpy()) < 2e-6
    assert rel_err(got[0].cpu().numpy(), gx64.cpu().numpy()) < 2e-6
    assert torch.allclose(got[0], x.grad, rtol=1e-5, atol=1e-5)
    assert rel_err(got[1].cpu().numpy(), lin.weight.grad.cpu().numpy()) < 2e-6
    assert rel_err(got[2].cpu().numpy(), lin.bias.grad.cpu().numpy()) < 2e-6
    # preconditions not met -> the layer itself
    with torch.no_grad():
        assert torch.equal(bracket_linear(x, lin), lin(x))
    odd = torch.nn.Linear(10, 6).cuda()
    z = torch.randn(5, 10).cuda().requires_grad_(True)
    bracket_linear(z, odd).sum().backward()
    assert z.grad is not None


@pytest.mark.parametrize("M,N,K,frac", [(600, 256, 256, 0.3), (6120, 256, 256, 0.05), (37, 12, 20, 0.5), (70, 68, 132, 1.0)])
def test_masked_wgrad_and_zero_rows(M, N, K, frac):
    """Rows flagged in the padding mask count as zero rows of grad_out; zero_masked_rows_ touches exactly them."""
    from uvhand_amd import _native
    g = torch.Generator().manual_seed(M + N)
    dy = torch.randn(M, N, generator=g).cuda()
    x = torch.randn(M, K, generator=g).cuda()
    mask = (torch.rand(M, generator=g) < frac).cuda()
    dym = dy.masked_fill(mask[:, None], 0.0)
    gw, gb = _native.linear_wgrad(dy, x, row_mask=mask)
    gw_ref, gb_ref = _native.linear_wgrad(dym, x)
    assert torch.equal(gw, gw_ref) and torch.equal(gb, gb_ref)         # same products, same order
    y = dy.clone()
    y[mask] = float("nan")                                             # masked rows may hold anything
    assert torch.equal(_native.zero_masked_rows_(y, mask), dym)
    with pytest.raises(RuntimeError, match="row_mask"):
        _native.linear_wgrad(dy, x, row_mask=mask[:-1])


def test_large_bracket_linear_keeps_the_vendor_gemm():
    """Beyond ~5000 rows x 256 features the forward and the input gradient stay on the vendor BLAS (its macro-tiles run 1.3x
    faster there): bit-identical to nn.Linear."""
    from uvhand_amd.functions.linear_func import bracket_linear
    torch.manual_seed(2)
    lin = torch.nn.Linear(256, 256).cuda()
    x = torch.randn(2, 3060, 256).cuda().requires_grad_(True)
    go = torch.randn(2, 3060, 256).cuda()
    y = bracket_linear(x, lin)
    y.backward(go)
    gx = x.grad.clone(); x.grad = None
    y_ref = lin(x)
    y_ref.backward(go)
    assert torch.equal(y, y_ref) and torch.equal(gx, x.grad)


def test_bracket_linear_masked_equals_linear_then_masked_fill():
    """value_proj + padding mask (modules/ms_deform_attn.py:96-98) through the masked-rows path."""
    from uvhand_amd.functions.linear_func import bracket_linear_masked
    torch.manual_seed(1)
    lin = torch.nn.Linear(256, 256).cuda()
    x = torch.randn(2, 85, 256).cuda().requires_grad_(True)
    mask = torch.zeros(2, 85, dtype=torch.bool)
    mask[0, 80:] = True; mask[1, 3] = True
    mask = mask.cuda()
    go = torch.randn(2, 85, 256).cuda()
    y = bracket_linear_masked(x, lin, mask)
    y.backward(go)
    got = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None; lin.zero_grad()
    y_ref = lin(x).masked_fill(mask[..., None], 0.0)
    y_ref.backward(go)
    y64 = (x.detach().double() @ lin.weight.detach().double().t() + lin.bias.detach().double()).masked_fill(mask[..., None], 0.0)
    assert rel_err(y.detach().cpu().numpy(), y64.cpu().numpy()) < 2e-6 and not y[mask].any()
    assert torch.allclose(got[0], x.grad, rtol=1e-5, atol=1e-5) and not got[0][mask].any()
    assert rel_err(got[1].cpu().numpy(), lin.weight.grad.cpu().numpy()) < 2e-6
    assert rel_err(got[2].cpu().numpy(), lin.bias.grad.cpu().numpy()) < 2e-6
    with torch.no_grad():                                              # preconditions not met -> the reference composition
        assert torch.equal(bracket_linear_masked(x, lin, mask), y_ref)


def _random_wgrad_shapes(count, seed):
    rng = np.random.RandomState(seed)
    return [(int(rng.choice([1, 2, 31, 64, 65, 257, 1000, 4097, 9000])), 4 * int(rng.randint(1, 100)), 4 * int(rng.randint(1, 100)),
             float(rng.choice([0.0, 0.0, 0.2, 1.0]))) for _ in range(count)]


@pytest.mark.parametrize("M,N,K,frac", _random_wgrad_shapes(20, 5))
def test_wgrad_random_shapes(M, N, K, frac):
    """Random row counts (1 ... 9000: one stage, ragged chunks, many splits) and feature counts (multiples of 4, ragged
    64-wide tiles), with and without a row mask, against fp64."""
    from uvhand_amd import _native
    g = torch.Generator().manual_seed(M * 7 + N + K)
    dy, x = torch.randn(M, N, generator=g).cuda(), torch.randn(M, K, generator=g).cuda()
    mask = (torch.rand(M, generator=g) < frac).cuda() if frac > 0 else None
    gw, gb = _native.linear_wgrad(dy, x, row_mask=mask)
    dym = dy if mask is None else dy.masked_fill(mask[:, None], 0.0)
    ref_w, ref_b = (dym.double().t() @ x.double()).cpu().numpy(), dym.double().sum(0).cpu().numpy()
    scale_w = max(1.0, float(np.abs(ref_w).max()))
    assert np.abs(gw.cpu().numpy() - ref_w).max() < 3e-6 * scale_w * max(1.0, np.sqrt(M) / 8)
    assert np.abs(gb.cpu().numpy() - ref_b).max() < 3e-6 * max(1.0, float(np.abs(ref_b).max())) * max(1.0, np.sqrt(M) / 8)


@pytest.mark.parametrize("M,N,K", [(600, 256, 256), (600, 384, 256), (33, 8, 12), (6120, 256, 1024), (33440, 1024, 256),
                                   (97, 72, 40), (4100, 64, 264)])
def test_bf16_operand_weight_gradient(M, N, K):
    """msda_linear_wgrad_masked_bf16: bf16 operands, fp32 products and sums.  With whole 8-column chunks (N, K multiples of 8)
    the products run on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, operands read from LDS with the transposing
    ds_read_b64_tr_b16): exact products, fp32 accumulation in another association than the fp32 kernel's — within 2e-6 of an
    fp64 product like that kernel, reproducible, and the masked rows of dY count as zero.  Other shapes widen the operands and
    take the fp32 kernel's chain: equal to it bit for bit."""
    from uvhand_amd import _native
    g = torch.Generator().manual_seed(M + N + K)
    go = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    gw, gb = _native.linear_wgrad(go, x)
    gw32, gb32 = _native.linear_wgrad(go.float(), x.float())
    assert gw.dtype == torch.float32
    ref, ref_b = go.double().t() @ x.double(), go.double().sum(0)
    tol = 2e-6 * max(1.0, (M ** 0.5) / 16)
    assert (gw.double() - ref).abs().max().item() < tol * ref.abs().max().item()
    assert (gb.double() - ref_b).abs().max().item() < tol * max(1.0, ref_b.abs().max().item())
    if N % 8 or K % 8:
        assert torch.equal(gw, gw32) and torch.equal(gb, gb32)
    else:
        assert (gw - gw32).abs().max().item() < 2 * tol * ref.abs().max().item()
        again = _native.linear_wgrad(go, x)
        assert torch.equal(gw, again[0]) and torch.equal(gb, again[1])               # fixed-order second stage: reproducible
        mask = torch.rand(M, generator=g).cuda() < 0.3
        mw, mb = _native.linear_wgrad(go, x, row_mask=mask)
        pw, pb = _native.linear_wgrad(go.masked_fill(mask[:, None], 0), x)
        assert torch.equal(mw, pw) and torch.equal(mb, pb)


def test_bracket_linear_under_bf16_autocast_uses_the_kernel_and_tracks_stock_autocast(monkeypatch):
    from uvhand_amd import _native
    from uvhand_amd.functions.linear_func import bracket_linear
    calls = []
    orig = _native.linear_wgrad
    monkeypatch.setattr(_native, "linear_wgrad", lambda *a, **k: (calls.append(a[0].dtype), orig(*a, **k))[1])
    torch.manual_seed(0)
    layer = torch.nn.Linear(256, 128).cuda()
    x = torch.randn(4, 150, 256, device="cuda", requires_grad=True)
    go = torch.randn(4, 150, 128, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = bracket_linear(x, layer)
    assert y.dtype == torch.bfloat16
    y.backward(go.to(y.dtype))
    got = (y.detach().float(), x.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone())
    assert calls == [torch.bfloat16] and got[2].dtype == torch.float32
    layer.zero_grad(); x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = layer(x)
    y2.backward(go.to(y2.dtype))
    assert torch.equal(got[0], y2.detach().float())                       # same forward
    for a, b in ((got[1], x.grad), (got[2], layer.weight.grad), (got[3], layer.bias.grad)):
        assert (a - b).abs().max().item() < 2e-2 * b.abs().max().item()   # stock rounds dW to bf16; the kernel does not


# ---- forward and input gradient of the same layers (msda_linear_forward_f32 / msda_linear_dgrad_f32) ----

def _random_rows_shapes(n, seed):
    rng = np.random.RandomState(seed)
    out = [(600, 256, 256, 0.0), (600, 384, 256, 0.0), (6120, 256, 256, 0.1), (33440, 256, 256, 0.0), (1, 4, 4, 0.0),
           (63, 64, 32, 0.5), (65, 68, 36, 0.0), (129, 132, 100, 1.0), (2, 1024, 256, 0.0), (300, 256, 1024, 0.2)]
    for _ in range(n):
        out.append((int(rng.randint(1, 700)), 4 * int(rng.randint(1, 100)), 4 * int(rng.randint(1, 100)), float(rng.choice([0.0, 0.3]))))
    return out


@pytest.mark.parametrize("rows,out_f,in_f,frac", _random_rows_shapes(12, 11))
def test_linear_forward_and_dgrad_match_fp64(rows, out_f, in_f, frac):
    """nn.Linear forward (modules/ms_deform_attn.py:96,100,101,139) and its input gradient, masked rows written as zeros
    (modules/ms_deform_attn.py:97-98), against the fp64 composition; exact-fp32 MFMA: tolerance of an fp32 dot product."""
    from uvhand_amd import _native
    g = torch.Generator().manual_seed(rows * 31 + out_f * 7 + in_f)
    x = torch.randn(rows, in_f, generator=g).cuda()
    w = (torch.randn(out_f, in_f, generator=g) * 0.1).cuda()
    b = torch.randn(out_f, generator=g).cuda()
    gy = torch.randn(rows, out_f, generator=g).cuda()
    mask = (torch.rand(rows, generator=g) < frac).cuda() if frac > 0 else None
    y = _native.linear_forward(x, w, b, mask)
    y_nb = _native.linear_forward(x, w, None, None)
    gx = _native.linear_dgrad(gy, w, mask)
    y64 = x.double() @ w.double().t() + b.double()
    gx64 = gy.double() @ w.double()
    if mask is not None:
        y64 = y64.masked_fill(mask[:, None], 0.0)
        gx64 = gx64.masked_fill(mask[:, None], 0.0)
        assert (y[mask] == 0).all() and (gx[mask] == 0).all()              # exact zeros, not small numbers
    assert rel_err(y.cpu().numpy(), y64.cpu().numpy()) < 2e-6
    assert rel_err(y_nb.cpu().numpy(), (x.double() @ w.double().t()).cpu().numpy()) < 2e-6
    assert rel_err(gx.cpu().numpy(), gx64.cpu().numpy()) < 2e-6
    # reproducible
    assert torch.equal(_native.linear_forward(x, w, b, mask), y) and torch.equal(_native.linear_dgrad(gy, w, mask), gx)


def test_linear_forward_handles_nan_free_padding_and_empty():
    """Rows past the matrix and columns past the weight never leak into the result (poisoned neighbours), rows = 0 is a no-op."""
    from uvhand_amd import _native
    big = torch.full((70 * 40 + 64,), float("nan"), device="cuda")
    x = big[32:32 + 70 * 40].view(70, 40)
    x.copy_(torch.randn(70, 40))
    wbig = torch.full((36 * 40 + 64,), float("nan"), device="cuda")
    w = wbig[32:32 + 36 * 40].view(36, 40)
    w.copy_(torch.randn(36, 40))
    y = _native.linear_forward(x, w)
    assert torch.isfinite(y).all()
    assert rel_err(y.cpu().numpy(), (x.double() @ w.double().t()).cpu().numpy()) < 2e-6
    gy = torch.randn(70, 36, device="cuda")
    assert torch.isfinite(_native.linear_dgrad(gy, w)).all()
    assert _native.linear_forward(x[:0], w).shape == (0, 36)
    with pytest.raises(RuntimeError):
        _native.linear_forward(torch.randn(8, 6, device="cuda"), torch.randn(8, 6, device="cuda"))     # 6 % 4 != 0
    with pytest.raises(RuntimeError):
        _native.linear_forward(x.double(), w.double())


def test_several_weight_gradients_in_one_call_equal_the_single_calls():
    """msda_linear_wgrad_multi_f32 (round 5: the module's three weight gradients, one second-stage launch for all): bitwise the
    results of the single calls — sizes of the module at the decoder shape (600 x 256 -> 256, 600 x 256 -> 768, 6120 x 256 -> 256
    with a row mask), plus a problem small enough to need no second stage."""
    import ctypes
    from uvhand_amd import _native
    lib = _native.load()
    g = torch.Generator().manual_seed(9)
    probs = [(600, 256, 256, False), (600, 768, 256, False), (6120, 256, 256, True), (64, 8, 4, False)]
    dys = [torch.randn(m, n, generator=g).cuda() for m, n, k, _ in probs]
    xs = [torch.randn(m, k, generator=g).cuda() for m, n, k, _ in probs]
    masks = [(torch.rand(m, generator=g) < 0.1).cuda() if mk else None for m, n, k, mk in probs]
    single = [_native.linear_wgrad(dy, x, want_bias=True, row_mask=mk) for dy, x, mk in zip(dys, xs, masks)]
    gws = [torch.empty(n, k, device="cuda") for m, n, k, _ in probs]
    gbs = [torch.empty(n, device="cuda") for m, n, k, _ in probs]
    lib.msda_linear_wgrad_workspace_bytes.restype = ctypes.c_ulonglong
    wss = [torch.empty(max(16, int(lib.msda_linear_wgrad_workspace_bytes(m, n, k))), dtype=torch.uint8, device="cuda") for m, n, k, _ in probs]
    VP, I = ctypes.c_void_p, ctypes.c_int
    arr = lambda ts: (VP * 4)(*[t.data_ptr() if t is not None else None for t in ts])
    ints = lambda vs: (I * 4)(*vs)
    lib.msda_linear_wgrad_multi_f32.restype = I
    lib.msda_linear_wgrad_multi_f32.argtypes = [I] + [VP] * 10
    rc = lib.msda_linear_wgrad_multi_f32(4, arr(dys), arr(xs), arr(masks), ints([p[0] for p in probs]), ints([p[1] for p in probs]),
                                         ints([p[2] for p in probs]), arr(gws), arr(gbs), arr(wss), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _native._lib.msda_last_error()
    torch.cuda.synchronize()
    for (gw1, gb1), gw, gb, pr in zip(single, gws, gbs, probs):
        assert torch.equal(gw, gw1) and torch.equal(gb, gb1), pr
    assert lib.msda_linear_wgrad_multi_f32(5, arr(dys), arr(xs), arr(masks), ints([1] * 4), ints([4] * 4), ints([4] * 4), arr(gws), arr(gbs),
                                           arr(wss), None) != 0                     # more than four problems: refused


def test_mixed_operand_weight_gradients_in_one_call_equal_the_single_calls():
    """msda_linear_wgrad_multi with operand types per problem (the module node under autocast: output_proj and value_proj on bf16
    operands, the merged projection on fp32): bitwise the results of msda_linear_wgrad_masked_bf16 / _f32 called one by one."""
    import ctypes
    from uvhand_amd import _native
    lib = _native.load()
    g = torch.Generator().manual_seed(10)
    probs = [(600, 256, 256, True), (600, 384, 256, False), (6120, 256, 256, True)]              # (M, N, K, bf16 operands)
    cast = lambda t, h: t.to(torch.bfloat16) if h else t
    dys = [cast(torch.randn(m, n, generator=g), h).cuda() for m, n, k, h in probs]
    xs = [cast(torch.randn(m, k, generator=g), h).cuda() for m, n, k, h in probs]
    single = [_native.linear_wgrad(dy, x, want_bias=True) for dy, x in zip(dys, xs)]
    gws = [torch.empty(n, k, device="cuda") for m, n, k, _ in probs]
    gbs = [torch.empty(n, device="cuda") for m, n, k, _ in probs]
    lib.msda_linear_wgrad_workspace_bytes.restype = ctypes.c_ulonglong
    wss = [torch.empty(max(16, int(lib.msda_linear_wgrad_workspace_bytes(m, n, k))), dtype=torch.uint8, device="cuda") for m, n, k, _ in probs]
    VP, I = ctypes.c_void_p, ctypes.c_int
    arr = lambda ts: (VP * 4)(*[t.data_ptr() for t in ts])
    ints = lambda vs: (I * 4)(*vs)
    lib.msda_linear_wgrad_multi.restype = I
    lib.msda_linear_wgrad_multi.argtypes = [I] + [VP] * 11
    rc = lib.msda_linear_wgrad_multi(3, arr(dys), arr(xs), ints([int(p[3]) for p in probs]), None, ints([p[0] for p in probs]),
                                     ints([p[1] for p in probs]), ints([p[2] for p in probs]), arr(gws), arr(gbs), arr(wss),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _native._lib.msda_last_error()
    torch.cuda.synchronize()
    for (gw1, gb1), gw, gb, pr in zip(single, gws, gbs, probs):
        assert torch.equal(gw, gw1) and torch.equal(gb, gb1), pr


@pytest.mark.parametrize("M,N,K,frac", [(33440, 256, 256, 0.0), (33440, 1024, 256, 0.05), (33440, 256, 1024, 0.0), (8200, 128, 384, 0.3),
                                        (9001, 384, 256, 0.0)])
def test_weight_gradient_at_the_training_shapes(M, N, K, frac):
    """The encoder layers' projections and FFN at the training shape (33 440 rows; N, K in 256 / 384 / 1024) against fp64, with and
    without a row mask, bias gradient included; reproducible run to run."""
    from uvhand_amd import _native
    g = torch.Generator().manual_seed(M + 3 * N + K)
    dy, x = torch.randn(M, N, generator=g).cuda(), torch.randn(M, K, generator=g).cuda()
    mask = (torch.rand(M, generator=g) < frac).cuda() if frac > 0 else None
    gw, gb = _native.linear_wgrad(dy, x, row_mask=mask)
    dym = dy if mask is None else dy.masked_fill(mask[:, None], 0.0)
    ref_w, ref_b = dym.double().t() @ x.double(), dym.double().sum(0)
    tol = 3e-6 * max(1.0, (M ** 0.5) / 8)
    assert (gw.double() - ref_w).abs().max().item() < tol * ref_w.abs().max().item()
    assert (gb.double() - ref_b).abs().max().item() < tol * max(1.0, ref_b.abs().max().item())
    again = _native.linear_wgrad(dy, x, row_mask=mask)
    assert torch.equal(gw, again[0]) and torch.equal(gb, again[1])
