"""Per-range point lists: on large problems the LDS-stage FORWARD leaves, for every (batch, head, level,
pixel range) of the backward's role-B plan, the indices of the sampling points whose taps may land in that range
(msda_forward_ws_* / msda_forward_prologue_ws_*, ListHeader in uvhand_amd/csrc/msda_d32.hip); role B of the same node's
backward reads its range's list instead of scanning all Lq*P points of the level once per range
(replaces the per-thread re-derivation of ms_deform_im2col_cuda.cuh:340-371).

Checked here: the lists are a duplicate-free SUPERSET of the exact answer (numpy restatement of the reference's tap rule,
ms_deform_im2col_cuda.cuh:285-288, :56-78), the backward from the lists equals the backward from a scan (grad_sampling_loc /
grad_attn_weight bit for bit, grad_value up to the order of a row's sum) and the C oracle at the bench's own sizes, a cleared
stamp or a foreign buffer falls back to the scan, and piled-up locations (more listed points than the LDS list holds) take
the chunked passes."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

C2 = [(48, 48), (24, 24), (12, 12), (6, 6)]
C4 = [(28, 28), (14, 14), (7, 7), (4, 4)]
BIG = {"cfg2_encoder": (2, 3060, C2), "cfg4_decoder": (32, 300, C4), "cfg4_encoder": (32, 1045, C4)}
M, P, D = 8, 4, 32
MAGIC = 0x4d53444c


@pytest.fixture(scope="module")
def native():
    from uvhand_amd import _native
    _native.load()
    return _native


@pytest.fixture(scope="module")
def oracle():
    from oracle import msda_oracle
    return msda_oracle


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
    """(header dict, counts[pairs*L*W, chunks], lists[pairs*L*W, chunks, cap]) of a list buffer (uint8 tensor)."""
    raw = table.cpu().numpy()
    hdr = raw[:64].view(np.int32)
    h = dict(zip(("magic", "W", "L", "chunks", "cap", "NP", "pairs", "qw"), hdr[:8].tolist()))
    n_sub = h["pairs"] * h["L"] * h["W"] * h["chunks"]
    counts_bytes = (n_sub * 2 + 63) & ~63
    counts = raw[64:64 + n_sub * 2].view(np.uint16).reshape(-1, h["chunks"])
    lists = raw[64 + counts_bytes:64 + counts_bytes + n_sub * h["cap"] * 2].view(np.uint16).reshape(-1, h["chunks"], h["cap"])
    return h, counts, lists


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
def test_plans_that_read_lists(native, name):
    """Lists exist exactly where role B is the kept-taps pass (more points per level than one pass sorts): the two encoder
    shapes; cfg-4 decoder (1200 points per level and pair, one range per level) scans once anyway and gets none."""
    N, Lq, shapes = BIG[name]
    S = sum(h * w for h, w in shapes)
    plan = native.describe_plan(N, S, M, D, len(shapes), Lq, P)
    nbytes = int(native._lib.msda_forward_workspace_bytes(N, S, M, D, len(shapes), Lq, P, 0))
    assert ("lists" in plan) == (name != "cfg4_decoder"), plan
    assert (nbytes > 0) == ("lists" in plan)
    assert "lists" not in native.describe_plan(N, S, M, D, len(shapes), Lq, P, deterministic=True)


@pytest.mark.parametrize("name", ["cfg2_encoder", "cfg4_encoder"])
def test_lists_are_a_duplicate_free_superset_of_the_exact_answer(native, name):
    sh, lsi, value, loc, attn, go = _case(name)
    N, Lq, shapes = BIG[name]
    out, table = native.ms_deform_attn_forward(value.cuda(), sh.cuda(), lsi.cuda(), loc.cuda(), attn.cuda(), 64, with_table=True)
    assert table is not None
    h, counts, lists = _decode(table)
    assert h["magic"] == MAGIC and h["L"] == 4 and h["NP"] == Lq * P and h["pairs"] == N * M and h["cap"] >= h["qw"] * P
    W, L, chunks = h["W"], h["L"], h["chunks"]
    assert (counts <= h["qw"] * P).all()
    exact = _exact_ranges(loc.numpy(), shapes, W)                            # [N, Lq, M, L, P, W]
    listed_total, exact_total = 0, 0
    for b in range(N):
        for m in range(M):
            for l in range(L):
                for t in range(W):
                    bucket = ((b * M + m) * L + l) * W + t
                    got = np.concatenate([lists[bucket, c, :counts[bucket, c]] for c in range(chunks)]).astype(np.int64)
                    assert len(np.unique(got)) == len(got), "a point listed twice"
                    assert (got < Lq * P).all()
                    # chunk c holds queries [c*qw, (c+1)*qw) only
                    for c in range(chunks):
                        q = lists[bucket, c, :counts[bucket, c]].astype(np.int64) // P
                        assert ((q >= c * h["qw"]) & (q < (c + 1) * h["qw"])).all()
                    want = np.nonzero(exact[b, :, m, l, :, t].reshape(-1))[0]      # index q*P + p
                    assert np.isin(want, got).all(), "a point with a tap in the range is missing from its list"
                    listed_total += len(got)
                    exact_total += len(want)
    assert listed_total <= 1.25 * exact_total + 64, (listed_total, exact_total)    # a superset, but a tight one


@pytest.mark.parametrize("rows", ["f32", "bf16"])
@pytest.mark.parametrize("name", list(BIG))
def test_backward_from_the_lists_equals_backward_from_a_scan(native, oracle, name, rows):
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
        assert name == "cfg4_decoder"
        return
    got = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=table, **kw)
    assert torch.equal(got[1], base[1]) and torch.equal(got[2], base[2])
    assert rel_err(got[0].cpu().numpy(), base[0].cpu().numpy()) < 2e-6
    r_gv, r_gl, r_ga = oracle.backward(go.numpy(), value.numpy(), sh.numpy(), lsi.numpy(), loc.numpy(), attn.numpy())
    assert rel_err(got[0].cpu().numpy(), r_gv) < 2e-5
    assert rel_err(got[2].cpu().numpy(), r_ga) < 2e-5
    # the deterministic flag never reads lists: bit-equal to its own result without the buffer
    det_a = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=table, deterministic=True, **kw)
    det_b = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, deterministic=True, **kw)
    assert all(torch.equal(x, y) for x, y in zip(det_a, det_b))


def test_a_cleared_stamp_or_a_foreign_buffer_means_a_scan(native):
    """Role B trusts the lists only behind the header the forward of the same plan wrote: with the stamp cleared (what a
    forward that could not write them does) or another plan's numbers in it, the result is the scan's; entries that are
    garbage behind a valid header still never index outside sampling_loc."""
    sh, lsi, value, loc, attn, go = _case("cfg2_encoder", seed=2)
    v, g, s, i, l, a = value.cuda(), go.cuda(), sh.cuda(), lsi.cuda(), loc.cuda(), attn.cuda()
    _, table = native.ms_deform_attn_forward(v, s, i, l, a, 64, with_table=True)
    base = native.ms_deform_attn_backward(v, s, i, l, a, g, 64)
    for word, val in ((0, 0), (1, 5), (3, 31), (4, 8), (5, 7)):              # magic, W, chunks, cap, NP
        t2 = table.clone()
        t2[:64].view(torch.int32)[word] = val
        got = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=t2)
        assert torch.equal(got[1], base[1]) and torch.equal(got[2], base[2])
        assert rel_err(got[0].cpu().numpy(), base[0].cpu().numpy()) < 2e-6
    t3 = table.clone()
    t3[64:] = 0xff                                                            # counts and entries all 0xffff
    got = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=t3)     # wrong numbers, but no fault
    torch.cuda.synchronize()
    assert torch.isfinite(got[0]).all()


def test_piled_up_locations_overflow_the_list_and_take_the_chunked_passes(native, oracle):
    """Every point within a few pixels: one range's list holds (almost) all Lq*P points of the level — more records than its
    workgroup's LDS holds — so that workgroup starts over in query chunks (which scan), the other ranges' lists are empty."""
    sh, lsi, value, loc, attn, go = _case("cfg2_encoder", seed=3)
    loc[..., 0] = 0.40 + 0.05 * loc[..., 0].clamp(0, 1)
    loc[..., 1] = 0.55 + 0.05 * loc[..., 1].clamp(0, 1)
    v, g, s, i, l, a = value.cuda(), go.cuda(), sh.cuda(), lsi.cuda(), loc.cuda(), attn.cuda()
    _, table = native.ms_deform_attn_forward(v, s, i, l, a, 64, with_table=True)
    h, counts, _ = _decode(table)
    per_bucket = counts.astype(np.int64).sum(1)
    assert per_bucket.max() >= 3060 * 4 * 0.9                                 # a list with nearly every point of the level
    got = native.ms_deform_attn_backward(v, s, i, l, a, g, 64, table=table)
    r_gv, r_gl, r_ga = oracle.backward(go.numpy(), value.numpy(), sh.numpy(), lsi.numpy(), loc.numpy(), attn.numpy())
    assert rel_err(got[0].cpu().numpy(), r_gv) < 2e-5
    assert rel_err(got[2].cpu().numpy(), r_ga) < 2e-5


def test_the_fused_prologue_pair_passes_lists_and_scratch_in_one_buffer(native):
    """msda_forward_prologue_ws_* + msda_backward_prologue_ws_f32 on a large problem: the buffer carries the lists first and
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
