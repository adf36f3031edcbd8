"""The decoder self-attention core (msda_attn32_*_f32, include/msda.h; SURVEY.md §8 f2, models/arctic_transformer.py:351,374-376)
against an fp64 restatement of softmax(q k^T * scale) v in torch — without dropout, and with the kernel's dropout mask restated in
numpy from the same seed (the hash is part of the ABI's contract between forward and backward) — and the drop-in wrapper against
nn.MultiheadAttention itself."""
import math

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _keep_mask(seed, pairs, Lq, Lk, p):
    """numpy restatement of at_hash / thresh (uvhand_amd/csrc/msda_attn.hip): True where the probability is kept."""
    if p == 0:
        return np.ones((pairs, Lq, Lk), bool)
    seed &= 0xFFFFFFFFFFFFFFFF
    lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32)
    q = np.arange(Lq, dtype=np.uint32)[None, :, None]
    key = np.arange(Lk, dtype=np.uint32)[None, None, :]
    pair = np.arange(pairs, dtype=np.uint32)[:, None, None]
    with np.errstate(over="ignore"):
        x = (q << np.uint32(16)) + key + (lo + pair * np.uint32(0x9E3779B1) + hi * np.uint32(0x85EBCA6B))
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x85EBCA6B)
        x = x ^ (x >> np.uint32(13)); x = x * np.uint32(0xC2B2AE35)
    thresh = np.uint32(min(4294967295.0, float(np.float32(p)) * 4294967296.0))
    return x >= thresh


def _reference(q, k, v, go, heads, scale, keep, p):
    """fp64: q [Lq, N, E], k / v [Lk, N, E] -> out, grads (keep: [N*heads, Lq, Lk] bool)."""
    Lq, N, E = q.shape
    Lk, hd = k.shape[0], E // heads
    qd, kd, vd = (t.double().detach().requires_grad_(True) for t in (q, k, v))
    split = lambda t, L: t.reshape(L, N * heads, hd).transpose(0, 1)
    s = torch.bmm(split(qd, Lq), split(kd, Lk).transpose(1, 2)) * scale
    pr = torch.softmax(s, -1) * torch.from_numpy(keep).to(s) / (1.0 - float(np.float32(p)))
    out = torch.bmm(pr, split(vd, Lk)).transpose(0, 1).reshape(Lq, N, E)
    out.backward(go.double())
    return out.detach(), qd.grad, kd.grad, vd.grad


CASES = [(300, 300, 4, 8, 0.0), (300, 300, 3, 8, 0.1), (37, 37, 2, 3, 0.0), (16, 320, 2, 1, 0.25), (320, 17, 1, 4, 0.5),
         (1, 1, 1, 1, 0.0), (129, 200, 2, 8, 0.1)]


@pytest.mark.parametrize("Lq,Lk,N,heads,p", CASES)
def test_attention_core_matches_fp64(Lq, Lk, N, heads, p):
    from uvhand_amd import _native
    _native.load()
    g = torch.Generator().manual_seed(Lq * 7 + Lk + N)
    E = heads * 32
    # q and k as column blocks of one packed tensor (what the wrapper passes), v and grad_out dense
    qk = torch.randn(max(Lq, Lk), N, 2 * E, generator=g).cuda()
    q, k = qk[:Lq, :, :E], qk[:Lk, :, E:]
    v, go = torch.randn(Lk, N, E, generator=g).cuda(), torch.randn(Lq, N, E, generator=g).cuda()
    scale = 1.0 / math.sqrt(32)
    seed_value = 0x1234567890ABCDEF ^ (Lq << 20)
    seed = torch.tensor([seed_value - (1 << 64) if seed_value >= (1 << 63) else seed_value], dtype=torch.int64).cuda()
    out, lse = _native.attn32_forward(q, k, v, heads, scale, p, seed if p > 0 else None)
    gq, gk, gv = _native.attn32_backward(q, k, v, out, lse, go, heads, scale, p, seed if p > 0 else None)
    torch.cuda.synchronize()
    keep = _keep_mask(seed_value, N * heads, Lq, Lk, p)
    if p > 0 and keep.size > 10000:
        assert abs(keep.mean() - (1 - p)) < 0.01
    r_out, r_gq, r_gk, r_gv = _reference(q.cpu(), k.cpu(), v.cpu(), go.cpu(), heads, scale, keep, p)
    # (relative to the tensor's largest entry; the floor matters for the single-key case only, whose exact dK / dQ are zero)
    rel = lambda a, b: ((a.cpu().double() - b).abs().max() / (b.abs().max() + 1e-1)).item()
    assert rel(out, r_out) < 5e-6
    assert rel(gq, r_gq) < 2e-5 and rel(gk, r_gk) < 2e-5 and rel(gv, r_gv) < 2e-5
    # log-sum-exp of the scaled scores
    s = torch.einsum("qbhd,kbhd->bhqk", q.cpu().double().view(Lq, N, heads, 32), k.cpu().double().view(Lk, N, heads, 32)) * scale
    assert (lse.cpu().double().view(N, heads, Lq) - torch.logsumexp(s, -1)).abs().max().item() < 1e-5
    # reproducible, and the backward writes into packed views
    out2, _ = _native.attn32_forward(q, k, v, heads, scale, p, seed if p > 0 else None)
    assert torch.equal(out, out2)
    if Lq == Lk:
        packed = torch.empty(Lq, N, 2 * E, device="cuda")
        _native.attn32_backward(q, k, v, out, lse, go, heads, scale, p, seed if p > 0 else None, grad_q=packed[..., :E], grad_k=packed[..., E:])
        assert torch.equal(packed[..., :E], gq) and torch.equal(packed[..., E:], gk)


def test_argument_errors():
    from uvhand_amd import _native
    _native.load()
    q = torch.randn(321, 1, 32, device="cuda")
    with pytest.raises(RuntimeError, match="320"):
        _native.attn32_forward(q, q, q, 1, 1.0)
    q = torch.randn(8, 1, 32, device="cuda")
    with pytest.raises(RuntimeError, match="seed"):
        _native.attn32_forward(q, q, q, 1, 1.0, 0.1, None)
    with pytest.raises(RuntimeError, match="contiguous last dimension"):
        _native.attn32_forward(q.transpose(0, 2).contiguous().transpose(0, 2), q, q, 1, 1.0)
    assert _native.attn32_supported(300, 300, 32) and not _native.attn32_supported(300, 300, 64) and not _native.attn32_supported(0, 3, 32)


@pytest.mark.parametrize("L,N", [(300, 4), (50, 2)])
def test_self_attention_wrapper_equals_the_module(L, N):
    """Eval mode (no dropout): the wrapper and nn.MultiheadAttention agree on the output and on every gradient."""
    from uvhand_amd.functions.attention_func import self_attention
    torch.manual_seed(3)
    mha = nn.MultiheadAttention(256, 8, dropout=0.1).cuda().eval()
    x_qk, x_v = torch.randn(L, N, 256, device="cuda"), torch.randn(L, N, 256, device="cuda")
    go = torch.randn(L, N, 256, device="cuda")
    res = []
    for fn in (lambda a, b: self_attention(mha, a, b), lambda a, b: mha(a, a, b)[0]):
        a, b = x_qk.clone().requires_grad_(True), x_v.clone().requires_grad_(True)
        mha.zero_grad()
        out = fn(a, b)
        out.backward(go)
        res.append([out.detach(), a.grad, b.grad] + [p.grad.clone() for p in mha.parameters()])
    for x, y in zip(*res):
        assert ((x - y).abs().max() / (y.abs().max() + 1e-30)).item() < 2e-5


def test_self_attention_wrapper_trains_with_dropout_and_falls_back():
    from uvhand_amd.functions import attention_func
    torch.manual_seed(5)
    mha = nn.MultiheadAttention(256, 8, dropout=0.1).cuda().train()
    x = torch.randn(300, 2, 256, device="cuda", requires_grad=True)
    torch.manual_seed(11)
    a = attention_func.self_attention(mha, x, x)
    torch.manual_seed(11)
    b = attention_func.self_attention(mha, x, x)
    assert torch.equal(a, b)                                        # the seed comes from torch's generator
    c = attention_func.self_attention(mha, x, x)
    assert not torch.equal(a, c)
    a.sum().backward()
    assert torch.isfinite(x.grad).all() and torch.isfinite(mha.in_proj_weight.grad).all()
    # what the kernels do not take goes to the module: 64-wide heads, too many queries
    wide = nn.MultiheadAttention(256, 4).cuda().eval()
    y = torch.randn(20, 2, 256, device="cuda")
    assert torch.allclose(attention_func.self_attention(wide, y, y), wide(y, y, y)[0], atol=1e-6)
    long = torch.randn(400, 1, 256, device="cuda")
    mha.eval()
    assert torch.allclose(attention_func.self_attention(mha, long, long), mha(long, long, long)[0], atol=1e-5)
