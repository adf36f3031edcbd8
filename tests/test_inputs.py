"""uvhand_amd.utils — the op's argument construction against a fixture produced by the
reference's own get_valid_ratio / get_reference_points (tests/golden/gen_golden_inputs.py), and the flatten /
decoder helpers against their definition (models/arctic_transformer.py:157-177, :413-419).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from uvhand_amd.utils import (decoder_reference_points, encoder_reference_points, flatten_feature_levels,
                              get_valid_ratio)


def _fixture():
    z = load_golden("inputs")
    shapes = [tuple(int(x) for x in hw) for hw in z["shapes"]]
    masks = [torch.from_numpy(z["mask%d" % i]) for i in range(len(shapes))]
    return z, shapes, masks


def test_valid_ratio_and_encoder_reference_points_match_the_reference():
    z, shapes, masks = _fixture()
    valid = torch.stack([get_valid_ratio(m) for m in masks], 1)
    assert np.array_equal(valid.numpy(), z["valid_ratios"])
    for spec in (shapes, torch.tensor(shapes, dtype=torch.long)):          # list of (H, W) or the int64 tensor
        ref = encoder_reference_points(spec, valid)
        assert ref.shape == z["enc_reference_points"].shape
        assert np.allclose(ref.numpy(), z["enc_reference_points"], rtol=0, atol=1e-7)


def test_flatten_feature_levels_layout():
    _, shapes, masks = _fixture()
    g = torch.Generator().manual_seed(3)
    N, C = masks[0].shape[0], 8
    srcs = [torch.randn(N, C, h, w, generator=g) for h, w in shapes]
    poss = [torch.randn(N, C, h, w, generator=g) for h, w in shapes]
    level_embed = torch.randn(len(shapes), C, generator=g)
    src, mask, pos, ss, lsi, valid = flatten_feature_levels(srcs, masks, poss, level_embed)
    S = sum(h * w for h, w in shapes)
    assert src.shape == (N, S, C) and pos.shape == (N, S, C) and mask.shape == (N, S)
    assert ss.dtype == torch.int64 and ss.tolist() == [list(s) for s in shapes]
    assert lsi.dtype == torch.int64 and lsi.tolist() == [0, 192, 240, 252]
    assert torch.equal(valid, torch.stack([get_valid_ratio(m) for m in masks], 1))
    # element (b, level l, pixel (y, x), channel c) sits at row level_start[l] + y*W + x
    for l, (h, w) in enumerate(shapes):
        y, x = h - 1, w // 2
        row = int(lsi[l]) + y * w + x
        assert torch.equal(src[:, row], srcs[l][:, :, y, x])
        assert torch.equal(pos[:, row], poss[l][:, :, y, x] + level_embed[l])
        assert torch.equal(mask[:, row], masks[l][:, y, x])


def test_decoder_reference_points_2d_and_42d():
    z, _, _ = _fixture()
    valid = torch.from_numpy(z["valid_ratios"])                            # [N, L, 2] = (w, h)
    N, L = valid.shape[:2]
    g = torch.Generator().manual_seed(5)
    p2 = torch.rand(N, 7, 2, generator=g)
    out2 = decoder_reference_points(p2, valid)
    assert out2.shape == (N, 7, L, 2)
    assert torch.equal(out2[1, 3, 2], p2[1, 3] * valid[1, 2])
    p42 = torch.rand(N, 7, 42, generator=g)
    out42 = decoder_reference_points(p42, valid)
    assert out42.shape == (N, 7, L, 42)
    assert torch.equal(out42[2, 5, 1, 0::2], p42[2, 5, 0::2] * valid[2, 1, 0])     # x coordinates scale with w
    assert torch.equal(out42[2, 5, 1, 1::2], p42[2, 5, 1::2] * valid[2, 1, 1])     # y coordinates with h
    with pytest.raises(ValueError):
        decoder_reference_points(torch.rand(N, 7, 4), valid)


# ---------------------------------------------------------------------------------------------
# callers.npz: the flatten block of DeformableTransformer.forward (models/arctic_transformer.py:157-177) and the
# per-layer reference points of DeformableTransformerDecoder.forward (:413-419), EXECUTED from the reference file
# by tests/golden/gen_golden_r02.py
# ---------------------------------------------------------------------------------------------
def _callers(device):
    z = load_golden("callers")
    L = z["shapes_in"].shape[0]
    to = lambda a: torch.from_numpy(a).to(device)
    srcs = [to(z["src%d" % i]) for i in range(L)]
    poss = [to(z["pos%d" % i]) for i in range(L)]
    masks = [to(z["mask%d" % i]) for i in range(L)]
    return z, srcs, masks, poss, to(z["level_embed"])


def _check_flatten(device):
    z, srcs, masks, poss, level_embed = _callers(device)
    src, mask, pos, ss, lsi, valid = flatten_feature_levels(srcs, masks, poss, level_embed)
    assert ss.device.type == device and ss.dtype == torch.int64 and lsi.dtype == torch.int64
    assert np.array_equal(src.cpu().numpy(), z["src_flatten"])
    assert np.array_equal(mask.cpu().numpy(), z["mask_flatten"])
    assert np.array_equal(pos.cpu().numpy(), z["lvl_pos_embed_flatten"])
    assert np.array_equal(ss.cpu().numpy(), z["spatial_shapes"])
    assert np.array_equal(lsi.cpu().numpy(), z["level_start_index"])
    # integer / copy outputs are exact everywhere; the fp32 divisions behind the valid ratios are correctly rounded
    # on the CPU (bit-equal to the reference's run) and within one ulp on the device
    same = np.array_equal if device == "cpu" else (lambda a, b: np.allclose(a, b, rtol=2e-7, atol=0))
    assert same(valid.cpu().numpy(), z["valid_ratios"])
    for width in (2, 42):
        out = decoder_reference_points(torch.from_numpy(z["dec_ref%d" % width]).to(device), valid)
        assert same(out.cpu().numpy(), z["dec_ref%d_input" % width])


def test_flatten_and_decoder_reference_points_match_the_reference():
    _check_flatten("cpu")


@pytest.mark.gpu
def test_flatten_and_decoder_reference_points_match_the_reference_on_device():
    _check_flatten("cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,shapes", [(3, 8, [(12, 16), (6, 8), (3, 4), (2, 2)]), (2, 256, [(28, 28), (14, 14), (7, 7), (4, 4)]),
                                         (1, 68, [(5, 67), (1, 1)]), (2, 4, [(1, 130)])])
def test_native_flatten_equals_the_reference_composition_with_gradients(N, C, shapes, monkeypatch):
    """msda_flatten_levels_f32 / msda_unflatten_levels_f32 against the reference's composition (flatten(2).transpose(1, 2),
    + level_embed, cat — models/arctic_transformer.py:162-173) run by PyTorch on the same device tensors: identical
    values (copies and one fp32 add), identical gradients for the feature maps, positional maps and the level embedding."""
    from uvhand_amd import _native
    calls = {"f": 0, "u": 0}
    for name, key in (("flatten_levels", "f"), ("unflatten_levels", "u")):
        orig = getattr(_native, name)
        monkeypatch.setattr(_native, name, lambda *a, _o=orig, _k=key, **k: (calls.__setitem__(_k, calls[_k] + 1), _o(*a, **k))[1])
    g = torch.Generator().manual_seed(N * 100 + C)
    mk = lambda: [torch.randn(N, C, h, w, generator=g).cuda().requires_grad_(True) for h, w in shapes]
    srcs, poss = mk(), mk()
    masks = [torch.zeros(N, h, w, dtype=torch.bool).cuda() for h, w in shapes]
    embed = torch.randn(len(shapes), C, generator=g).cuda().requires_grad_(True)
    src, mask, pos, ss, lsi, valid = flatten_feature_levels(srcs, masks, poss, embed)
    assert calls["f"] == 1
    S = sum(h * w for h, w in shapes)
    ref_src = torch.cat([t.flatten(2).transpose(1, 2) for t in srcs], 1)
    ref_pos = torch.cat([t.flatten(2).transpose(1, 2) + embed[l].view(1, 1, -1) for l, t in enumerate(poss)], 1)
    assert src.shape == (N, S, C) and torch.equal(src, ref_src) and torch.equal(pos, ref_pos)
    assert ss.tolist() == [list(s) for s in shapes] and int(lsi[-1]) == S - shapes[-1][0] * shapes[-1][1]
    g1, g2 = torch.randn(N, S, C, generator=g).cuda(), torch.randn(N, S, C, generator=g).cuda()
    got = torch.autograd.grad([src, pos], srcs + poss + [embed], [g1, g2])
    want = torch.autograd.grad([ref_src, ref_pos], srcs + poss + [embed], [g1, g2])
    assert calls["u"] == 1
    for a, b in zip(got[:-1], want[:-1]):
        assert torch.equal(a, b)
    assert torch.allclose(got[-1], want[-1], rtol=1e-5, atol=1e-5)            # level-embedding column sums: summation order
