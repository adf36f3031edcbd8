"""CPU-side checks of the f2 layers: names, state_dict keys and seeded construction follow the reference's classes
(models/arctic_transformer.py:261-300, :334-391; fixtures from tests/golden/gen_golden_r02.py), and the fused add+LayerNorm
wrapper falls back to the framework's layers where the kernel does not apply."""
import torch
from torch import nn

from conftest import load_golden


def test_layer_state_dict_keys_match_the_reference():
    from uvhand_amd.modules import DeformableTransformerDecoderLayer, DeformableTransformerEncoderLayer
    for cls, fixture in ((DeformableTransformerEncoderLayer, "layer_encoder"), (DeformableTransformerDecoderLayer, "layer_decoder_2d")):
        z = load_golden(fixture)
        ref_keys = [k[len("state."):] for k in z if k.startswith("state.")]
        layer = cls(64, 128, 0.0, "relu", 4, 2, 4)
        assert list(layer.state_dict().keys()) == ref_keys
        for k, v in layer.state_dict().items():
            assert tuple(v.shape) == z["state." + k].shape, k
        layer.load_state_dict({k: torch.from_numpy(z["state." + k]) for k in ref_keys}, strict=True)


def test_decoder_layer_module_names():
    from uvhand_amd.modules import DeformableTransformerDecoderLayer
    layer = DeformableTransformerDecoderLayer()
    names = [n for n, _ in layer.named_children()]
    assert names == ["cross_attn", "dropout1", "norm1", "self_attn", "dropout2", "norm2", "linear1", "dropout3", "linear2",
                     "dropout4", "norm3", "inter_rp", "attn_matrix"]
    assert isinstance(layer.self_attn, nn.MultiheadAttention) and layer.self_attn.num_heads == 8
    assert layer.linear1.out_features == 1024 and layer.dropout1.p == 0.1


def test_add_layer_norm_falls_back_on_cpu():
    from uvhand_amd.functions.layernorm_func import add_layer_norm
    norm = nn.LayerNorm(12)
    x, r = torch.randn(5, 12), torch.randn(5, 12)
    assert torch.equal(add_layer_norm(x, r, norm), norm(x + r))
    assert torch.equal(add_layer_norm(x, None, norm), norm(x))


def test_decoder_layer_computes_the_attention_matrix_only_for_a_listener():
    """The reference's decoder layer passes the head-averaged self-attention matrix through a parameter-free `attn_matrix` module
    (a tap for hooks, models/arctic_transformer.py:374-378).  Here the matrix is computed when a hook on that module listens — or
    always with `always_attention_matrix` — and skipped otherwise (nn.MultiheadAttention's fused path); in eval mode the layer's
    output is the same either way."""
    from uvhand_amd.modules import DeformableTransformerDecoderLayer

    class NoSampling(nn.Module):                   # the sampling module needs the GPU library: not this test's subject
        def forward(self, query, *args):
            return query * 0.5

    torch.manual_seed(0)
    layer = DeformableTransformerDecoderLayer(32, 64, 0.1, "relu", 2, 4, 2)
    layer.cross_attn = NoSampling()
    layer.eval()
    tgt, pos, ref = torch.randn(2, 7, 32), torch.randn(2, 7, 32), torch.rand(2, 7, 2, 2)
    args = (tgt, pos, ref, torch.randn(2, 20, 32), torch.tensor([[4, 4], [2, 2]]), torch.tensor([0, 16]))
    assert layer.always_attention_matrix is False
    calls = []
    real = layer.self_attn.forward
    layer.self_attn.forward = lambda *a, **k: (calls.append(k.get("need_weights")), real(*a, **k))[1]
    plain = layer(*args)
    seen = []
    handle = layer.attn_matrix.register_forward_hook(lambda mod, inp, out: seen.append(inp[0]))
    hooked = layer(*args)
    handle.remove()
    layer.always_attention_matrix = True
    always = layer(*args)
    assert calls == [False, True, True]
    # hooks registered for EVERY module and backward hooks on the tap count as listeners too
    calls.clear()
    layer.always_attention_matrix = False
    g = torch.nn.modules.module.register_module_forward_hook(lambda mod, inp, out: None)
    layer(*args)
    g.remove()
    b = layer.attn_matrix.register_full_backward_hook(lambda mod, gi, go: None)
    layer(*args)
    b.remove()
    layer(*args)
    assert calls == [True, True, False]
    assert len(seen) == 1 and seen[0].shape == (2, 7, 7) and torch.allclose(seen[0].sum(-1), torch.ones(2, 7), atol=1e-5)
    assert torch.allclose(plain, hooked, atol=1e-5) and torch.equal(hooked, always)
