"""Per-point range masks: on large problems the LDS-stage FORWARD leaves one byte per sampling point —
bit t set iff a tap of the point may land in pixel range t of its level, the ranges being those of the backward's role-B plan
(msda_forward_ws_* / msda_forward_prologue_ws_*, MaskHeader in uvhand_amd/csrc/msda_d32.hip); role B of the same node's
backward finds its candidates with a coalesced byte scan instead of a strided float scan of sampling_loc once per range
(replaces the per-thread re-derivation of ms_deform_im2col_cuda.cuh:340-371).

Checked here: the masks are a tight SUPERSET of the exact answer (numpy restatement of the reference's tap rule,
ms_deform_im2col_cuda.cuh:285-288, :56-78), the backward from the masks equals the backward from a scan (grad_sampling_loc /
grad_attn_weight bit for bit, grad_value up to the order of a row's sum) and the C oracle at the bench's own sizes, a cleared
stamp or a foreign buffer falls back to the scan, and piled-up locations (more candidates than the LDS list holds) take
the chunked passes."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

C2 = [(48, 48), (24, 24), (12, 12), (6, 6)]
C4 = [(28, 28), (14, 14), (7, 7), (4, 4)]
BIG = {"cfg2_encoder": (2, 3060, C2), "cfg4_decoder": (32, 300, C4), "cfg4_encoder": (32, 1045, C4),
       "l3_w4_encoder": (4, 2100, [(40, 40), (20, 20), (10, 10)])}      # a 3-level pyramid whose plan has W = 4 ranges per level
M, P, D = 8, 4, 32
MAGIC = 0x4d53444d


@pytest.fixture(scope="module")
def native():
    from uvhand_amd import _native
    _native.load()
    return _native


def _case(name, seed=0, spread=1.3, shift=-0.15):
    N, Lq, shapes = BIG[name]
    g = torch.Generator().manual_seed(4321 + seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    sh = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    value = torch.rand(N, S, M, D, generator=g) - 0.5
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * spread + shift      # some points outside the maps
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = attn / attn.sum((-1, -2), keepdim=True)
    go = torch.randn(N, Lq, M * D, generator=g)
    return sh, lsi, value, loc, attn, go


def _decode(table):
    """(header dict, masks[pairs, L, NP]) of a mask buffer (uint8 tensor)."""
    raw = table.cpu().numpy()
    h = dict(zip(("magic", "W", "L", "NP", "pairs"), raw[:64].view(np.int32)[:5].tolist()))
    n = h["pairs"] * h["L"] * h["NP"]
    return h, raw[64:64 + n].reshape(h["pairs"], h["L"], h["NP"])


def _exact_ranges(loc, shapes, W):
    """For every point, the set of ranges (of its level) that one of its VALID taps lands in, by the reference's rule:
    h_im = y*H - 0.5 in (-1, H), taps (floor, floor + 1) inside the map.  Returns bool [N, Lq, M, L, P, W]."""
    N, Lq, Mh, L, Pp, _ = loc.shape
    out = np.zeros((N, Lq, Mh, L, Pp, W), bool)
    for l, (H, Wd) in enumerate(shapes):
        x, y = loc[:, :, :, l, :, 0].astype(np.float32), loc[:, :, :, l, :, 1].astype(np.float32)
        h_im, w_im = y * np.float32(H) - np.float32(0.5), x * np.float32(Wd) - np.float32(0.5)
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < Wd)
        h0, w0 = np.floor(h_im).astype(np.int64), np.floor(w_im).astype(np.int64)
        bounds = np.array([(t * H * Wd) // W for t in range(W + 1)])
        for dh in (0, 1):
            for dw in (0, 1):
                hh, ww = h0 + dh, w0 + dw
                ok = inside & (hh >= 0) & (hh < H) & (ww >= 0) & (ww < Wd)
                pix = np.where(ok, hh * Wd + ww, 0)
                t = np.searchsorted(bounds, pix, side="right") - 1        # range t holds [bounds[t], bounds[t+1])
                for tt in range(W):
                    out[:, :, :, l, :, tt] |= ok & (t == tt)
    return out


@pytest.mark.parametrize("name", list(BIG))
def test_plans_that_read_masks(native, name):
    """Masks exist where role B is the kept-taps pass (more points per level than one pass sorts) with at least four ranges per
    level: cfg-2 encoder (W = 6).  cfg-4 encoder (W = 2: its scan reads every point twice, cheaper than the forward's bytes —
    measured, profiles/r05_notes.md) and cfg-4 decoder (one range per level, one pass) get none."""
    N, Lq, shapes = BIG[name]
    S = sum(h * w for h, w in shapes)
    plan = native.describe_plan(N, S, M, D, len(shapes), Lq, P)
    nbytes = int(native._lib.msda_forward_workspace_bytes(N, S, M, D, len(shapes), Lq, P, 0))
    assert ("masks" in plan) == (name in ("cfg2_encoder", "l3_w4_encoder")), plan
    assert nbytes == (64 + N * M * len(shapes) * Lq * P if "masks" in plan else 0)
    assert "masks" not in native.describe_plan(N, S, M, D, len(shapes), Lq, P, deterministic=True)




@pytest.mark.parametrize("name", ["cfg2_encoder", "l3_w4_encoder"])
def test_masks_are_a_tight_superset_of_the_exact_answer(native, name):
    sh, lsi, value, loc, attn, go = _case(name)
    N, Lq, shapes = BIG[name]
    out, table = native.ms_deform_attn_forward(value.cuda(), sh.cuda(), lsi.cuda(), loc.cuda(), attn.cuda(), 64, with_table=True)
    assert table is not None
    h, masks = _decode(table)
    assert h["magic"] == MAGIC and h["L"] == len(shapes) and h["NP"] == Lq * P and h["pairs"] == N * M and 1 <= h["W"] <= 8
    W = h["W"]
    exact = _exact_ranges(loc.numpy(), shapes, W)                            # [N, Lq, M, L, P, W]
    got = ((masks[..., None] >> np.arange(W)) & 1).astype(bool)              # [pairs, L, NP, W]
    got = got.reshape(N, M, len(shapes), Lq, P, W).transpose(0, 3, 1, 2, 4, 5)   # -> [N, Lq, M, L, P, W]
    assert (masks >> W == 0).all(), "bits beyond the plan's ranges"
    assert not (exact & ~got).any(), "a point with a tap in a range whose bit is clear"
    assert got.sum() <= 1.25 * exact.sum() + 64, (int(got.sum()), int(exact.sum()))       # a superset, but a tight one
    outside = ~exact.any(-1)                                                  # points with no valid tap at all ...
    far = outside & ((loc.numpy() < -0.1) | (loc.numpy() > 1.1)).any(-1)       # ... that lie well outside the map
    assert not got[far].any()


@pytest.mark.parametrize("rows", ["f32", "bf16"])
@pytest.mark.parametrize("name", list(BIG))
def test_backward_from_the_masks_equals_backward_from_a_scan(native, oracle, name, rows):
    sh, lsi, value, loc, attn, go = _case(name, seed=1)
    bf16 = rows == "bf16"
    dt = torch.bfloat16 if bf16 else torch.float32
    if bf16:
        value, go = value.to(dt).float(), go.to(dt).float()
    v, g = value.cuda().to(dt), go.cuda().to(dt)
    s, i, l, a = sh.cuda(), lsi.cuda(), loc.cuda(), attn.cuda()
    out_plain = native.ms_deform_attn_forward(v, s, i, l, a, 64)
    out, table = native.ms_deform_attn_forward(v, s, i, l, a, 64, with_table=True)
    assert torch.equal(out, out_plain)
    kw = {"fp32_grad_value": True} if bf16 else {}
    base = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, **kw)
    if table is None:
        assert name in ("cfg4_decoder", "cfg4_encoder")
        return
    got = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=table, **kw)
    assert torch.equal(got[1], base[1]) and torch.equal(got[2], base[2])
    assert rel_err(got[0].cpu().numpy(), base[0].cpu().numpy()) < 2e-6
    r_gv, r_gl, r_ga = oracle.backward(go.numpy(), value.numpy(), sh.numpy(), lsi.numpy(), loc.numpy(), attn.numpy())
    assert rel_err(got[0].cpu().numpy(), r_gv) < 2e-5
    assert rel_err(got[2].cpu().numpy(), r_ga) < 2e-5
    # the deterministic flag never reads masks: bit-equal to its own result without the buffer
    det_a = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=table, deterministic=True, **kw)
    det_b = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, deterministic=True, **kw)
    assert all(torch.equal(x, y) for x, y in zip(det_a, det_b))


def test_a_cleared_stamp_or_a_foreign_buffer_means_a_scan(native):
    """Role B trusts the masks only behind the header the forward of the same plan wrote: with the stamp cleared (what a
    forward that could not write them does) or another plan's numbers in it, the result is the scan's; masks that are
    garbage behind a valid header give wrong candidates at worst (all bits set = every point a candidate: still exact)."""
    sh, lsi, value, loc, attn, go = _case("cfg2_encoder", seed=2)
    v, g, s, i, l, a = value.cuda(), go.cuda(), sh.cuda(), lsi.cuda(), loc.cuda(), attn.cuda()
    _, table = native.ms_deform_attn_forward(v, s, i, l, a, 64, with_table=True)
    base = native.ms_deform_attn_backward(v, s, i, l, a, g, 64)
    for word, val in ((0, 0), (1, 5), (2, 3), (3, 7), (4, 1)):               # magic, W, L, NP, pairs
        t2 = table.clone()
        t2[:64].view(torch.int32)[word] = val
        got = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=t2)
        assert torch.equal(got[1], base[1]) and torch.equal(got[2], base[2])
        assert rel_err(got[0].cpu().numpy(), base[0].cpu().numpy()) < 2e-6
    t3 = table.clone()
    t3[64:] = 0xff                                                            # every point a candidate of every range
    got = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=t3)
    assert torch.equal(got[1], base[1]) and torch.equal(got[2], base[2])
    assert rel_err(got[0].cpu().numpy(), base[0].cpu().numpy()) < 2e-6


def test_piled_up_locations_overflow_the_list_and_take_the_chunked_passes(native, oracle):
    """Every point within a few pixels: one range's workgroup finds (almost) all Lq*P points of the level to be candidates —
    more records than its LDS holds — and starts over in query chunks (which scan); the other ranges find none."""
    sh, lsi, value, loc, attn, go = _case("cfg2_encoder", seed=3)
    loc[..., 0] = 0.40 + 0.05 * loc[..., 0].clamp(0, 1)
    loc[..., 1] = 0.55 + 0.05 * loc[..., 1].clamp(0, 1)
    v, g, s, i, l, a = value.cuda(), go.cuda(), sh.cuda(), lsi.cuda(), loc.cuda(), attn.cuda()
    _, table = native.ms_deform_attn_forward(v, s, i, l, a, 64, with_table=True)
    h, masks = _decode(table)
    per_range = ((masks[..., None] >> np.arange(h["W"])) & 1).sum(2)          # [pairs, L, W]
    assert per_range.max() >= 3060 * 4 * 0.9                                 # a range with nearly every point of the level
    got = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=table)
    r_gv, r_gl, r_ga = oracle.backward(go.numpy(), value.numpy(), sh.numpy(), lsi.numpy(), loc.numpy(), attn.numpy())
    assert rel_err(got[0].cpu().numpy(), r_gv) < 2e-5
    assert rel_err(got[2].cpu().numpy(), r_ga) < 2e-5


def test_the_fused_prologue_pair_passes_masks_and_scratch_in_one_buffer(native):
    """msda_forward_prologue_ws_* + msda_backward_prologue_ws_f32 on a large problem: the buffer carries the masks first and
    the per-head reference-point scratch behind them (msda_backward_workspace_bytes with MSDA_FLAG_FORWARD_TABLE)."""
    name = "cfg2_encoder"
    N, Lq, shapes = BIG[name]
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(77)
    sh = torch.tensor(shapes, dtype=torch.long).cuda()
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    value = (torch.rand(N, S, M, D, generator=g) - 0.5).cuda()
    ref = (torch.rand(N, Lq, L, 2, generator=g) * 1.2 - 0.1).cuda()
    off = (torch.randn(N, Lq, M, L, P, 2, generator=g) * 2.0).cuda()
    logits = torch.randn(N, Lq, M, L * P, generator=g).cuda()
    go = torch.randn(N, Lq, M * D, generator=g).cuda()
    lib = native._lib
    fw = int(lib.msda_forward_workspace_bytes(N, S, M, D, L, Lq, P, native.FLAG_PROLOGUE))
    both = int(lib.msda_backward_workspace_bytes(N, S, M, D, L, Lq, P, native.FLAG_PROLOGUE | native.FLAG_FORWARD_TABLE))
    heads = int(lib.msda_backward_workspace_bytes(N, S, M, D, L, Lq, P, native.FLAG_PROLOGUE))
    assert fw > 0 and heads > 0 and both == ((fw + 255) & ~255) + heads
    out, loc, attn, table = native.ms_deform_attn_forward_prologue(value, sh, lsi, ref, off, logits, 64, with_table=True)
    assert table is not None and table.numel() == both
    base = native.ms_deform_attn_backward_prologue(value, sh, lsi, loc, attn, go)
    got = native.ms_deform_attn_backward_prologue(value, sh, lsi, loc, attn, go, table=table)
    for k in (1, 2, 3):                                                       # offsets, logits, reference points: bit for bit
        assert torch.equal(got[k], base[k]), k
    assert rel_err(got[0].cpu().numpy(), base[0].cpu().numpy()) < 2e-6
