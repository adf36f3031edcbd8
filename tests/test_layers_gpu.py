"""SURVEY.md §8 f2 — the encoder / decoder layers around the op, against fixtures produced by executing the reference's
own class definitions (models/arctic_transformer.py:261-300, :334-391; tests/golden/gen_golden_r02.py), and the fused
residual-add + LayerNorm kernel against an fp64 composition."""
import numpy as np
import pytest
import torch
from torch import nn

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _cuda(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.requires_grad_(True) if grad else t


def _load(layer, z):
    state = {k[len("state."):]: torch.from_numpy(v) for k, v in z.items() if k.startswith("state.")}
    missing, unexpected = layer.load_state_dict(state, strict=True)
    assert not missing and not unexpected
    return layer.cuda()


def _check_param_grads(layer, z, tol=3e-4):
    for name, p in layer.named_parameters():
        assert p.grad is not None, name
        assert rel_err(p.grad.cpu().numpy(), z["pgrad." + name]) < tol, name


@pytest.mark.parametrize("train", [False, True])
def test_encoder_layer_matches_reference(train):
    from uvhand_amd.modules import DeformableTransformerEncoderLayer
    z = load_golden("layer_encoder")
    layer = _load(DeformableTransformerEncoderLayer(64, 128, 0.0, "relu", 4, 2, 4), z)
    layer.train(train)                                              # dropout = 0: both modes must reproduce the fixture
    src, pos = _cuda(z["src"], True), _cuda(z["pos"], True)
    out = layer(src, pos, _cuda(z["ref"]), _cuda(z["shapes"]), _cuda(z["level_start"]), _cuda(z["mask"]))
    out.backward(_cuda(z["gout"]))
    torch.cuda.synchronize()
    assert rel_err(out.detach().cpu().numpy(), z["out"]) < 1e-4
    assert rel_err(src.grad.cpu().numpy(), z["grad_src"]) < 2e-4
    assert rel_err(pos.grad.cpu().numpy(), z["grad_pos"]) < 2e-4
    _check_param_grads(layer, z)


@pytest.mark.parametrize("width", [2, 42])
def test_decoder_layer_matches_reference(width):
    from uvhand_amd.modules import DeformableTransformerDecoderLayer
    z = load_golden("layer_decoder_%dd" % width)
    layer = _load(DeformableTransformerDecoderLayer(64, 128, 0.0, "relu", 4, 2, 4), z)
    tgt, qpos, memory = _cuda(z["tgt"], True), _cuda(z["qpos"], True), _cuda(z["memory"], True)
    out = layer(tgt, qpos, _cuda(z["ref"]), memory, _cuda(z["shapes"]), _cuda(z["level_start"]), _cuda(z["mask"]))
    out.backward(_cuda(z["gout"]))
    torch.cuda.synchronize()
    assert rel_err(out.detach().cpu().numpy(), z["out"]) < 1e-4
    assert rel_err(tgt.grad.cpu().numpy(), z["grad_tgt"]) < 2e-4
    assert rel_err(qpos.grad.cpu().numpy(), z["grad_qpos"]) < 2e-4
    assert rel_err(memory.grad.cpu().numpy(), z["grad_memory"]) < 2e-4
    _check_param_grads(layer, z)


def test_layers_use_the_fused_kernels(monkeypatch):
    """The product path is the HIP one: the add+LayerNorm kernel and the weight-gradient kernel really run."""
    from uvhand_amd import _native
    from uvhand_amd.modules import DeformableTransformerEncoderLayer
    calls = {"ln_f": 0, "ln_b": 0, "wgrad": 0}
    for name, key in (("add_layernorm_forward", "ln_f"), ("add_layernorm_backward", "ln_b"), ("linear_wgrad", "wgrad")):
        orig = getattr(_native, name)

        def wrapped(*a, _o=orig, _k=key, **kw):
            calls[_k] += 1
            return _o(*a, **kw)
        monkeypatch.setattr(_native, name, wrapped)
    z = load_golden("layer_encoder")
    layer = _load(DeformableTransformerEncoderLayer(64, 128, 0.0, "relu", 4, 2, 4), z)
    layer.self_attn.cpp_node = False          # count the attention module's kernels through the Python composition
    out = layer(_cuda(z["src"], True), _cuda(z["pos"]), _cuda(z["ref"]), _cuda(z["shapes"]), _cuda(z["level_start"]))
    out.sum().backward()
    assert calls["ln_f"] == 2 and calls["ln_b"] == 2 and calls["wgrad"] >= 4
    # by default the attention module is ONE C++ node (its three weight gradients are queued from there): the FFN's two remain
    calls.update(ln_f=0, ln_b=0, wgrad=0)
    layer.self_attn.cpp_node = True
    out = layer(_cuda(z["src"], True), _cuda(z["pos"]), _cuda(z["ref"]), _cuda(z["shapes"]), _cuda(z["level_start"]))
    out.sum().backward()
    from uvhand_amd import _ext
    if _ext.get() is not None and hasattr(_ext.get(), "module_forward"):
        assert calls["ln_f"] == 2 and calls["ln_b"] == 2 and calls["wgrad"] == 2


def test_dropout_in_training_keeps_torch_random_stream():
    """With p > 0 the masks are nn.Dropout's: same seed -> same output as the unfused composition of the same modules."""
    from uvhand_amd.modules import DeformableTransformerEncoderLayer
    z = load_golden("layer_encoder")
    layer = _load(DeformableTransformerEncoderLayer(64, 128, 0.3, "relu", 4, 2, 4), z).train()
    args = (_cuda(z["src"]), _cuda(z["pos"]), _cuda(z["ref"]), _cuda(z["shapes"]), _cuda(z["level_start"]))
    torch.manual_seed(5)
    fused = layer(*args)
    torch.manual_seed(5)
    src, pos = args[0], args[1]
    src2 = layer.self_attn(src + pos, args[2], src, args[3], args[4], None)
    s = layer.norm1(src + layer.dropout1(src2))
    s2 = layer.linear2(layer.dropout2(torch.relu(layer.linear1(s))))
    plain = layer.norm2(s + layer.dropout3(s2))
    assert rel_err(fused.detach().cpu().numpy(), plain.detach().cpu().numpy()) < 1e-5


@pytest.mark.parametrize("rows,d,with_res", [(1, 256, True), (7, 64, True), (600, 256, True), (6120, 256, False),
                                             (33, 512, True), (129, 1024, True), (50, 260, True)])
def test_add_layernorm_kernel_matches_fp64(rows, d, with_res):
    from uvhand_amd.functions.layernorm_func import add_layer_norm
    g = torch.Generator().manual_seed(rows * 7 + d)
    norm = nn.LayerNorm(d).cuda()
    with torch.no_grad():
        norm.weight.copy_(torch.randn(d, generator=g) * 0.5 + 1.0)
        norm.bias.copy_(torch.randn(d, generator=g) * 0.3)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.7).cuda().requires_grad_(True)
    r = (torch.randn(rows, d, generator=g)).cuda().requires_grad_(True) if with_res else None
    gy = torch.randn(rows, d, generator=g).cuda()
    y = add_layer_norm(x, r, norm)
    y.backward(gy)
    # fp64 composition of the same two layers
    x64 = x.detach().double().requires_grad_(True)
    r64 = r.detach().double().requires_grad_(True) if with_res else None
    w64, b64 = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    y64 = torch.nn.functional.layer_norm(x64 if r64 is None else x64 + r64, (d,), w64, b64, norm.eps)
    y64.backward(gy.double())
    assert rel_err(y.detach().cpu().numpy(), y64.detach().cpu().numpy()) < 2e-6
    assert rel_err(x.grad.cpu().numpy(), x64.grad.cpu().numpy()) < 5e-6
    if with_res:
        assert torch.equal(x.grad, r.grad)
    assert rel_err(norm.weight.grad.cpu().numpy(), w64.grad.cpu().numpy()) < 5e-6
    assert rel_err(norm.bias.grad.cpu().numpy(), b64.grad.cpu().numpy()) < 5e-6
    # reproducible: fixed-order partial sums
    gw1 = norm.weight.grad.clone()
    norm.zero_grad(); x.grad = None
    add_layer_norm(x, r, norm).backward(gy)
    assert torch.equal(norm.weight.grad, gw1)


@pytest.mark.parametrize("rows,d,ffn,p", [(600, 256, 1024, 0.0), (6120, 256, 1024, 0.1), (33, 64, 128, 0.3), (1, 32, 64, 0.5)])
def test_fused_ffn_equals_the_composition_of_the_same_modules(rows, d, ffn, p):
    """linear2(dropout(relu(linear1(x)))) as one autograd node (models/arctic_transformer.py:283-287): the
    same values and the same gradients — input, both weights, both biases — as the reference's composition of nn.Linear / relu /
    nn.Dropout run from the same seed (the node calls PyTorch's own dropout kernel, so the masks are identical), and the random
    stream is left where nn.Dropout leaves it."""
    from uvhand_amd.functions.linear_func import _FusedFFNFn, fused_ffn
    g = torch.Generator().manual_seed(rows + d)
    l1, l2, drop = nn.Linear(d, ffn).cuda(), nn.Linear(ffn, d).cuda(), nn.Dropout(p).train()
    x = torch.randn(2, rows, d, generator=g).cuda().requires_grad_(True) if rows > 1 else torch.randn(rows, d, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(x.shape, generator=g).cuda()
    torch.manual_seed(11)
    y = fused_ffn(x, l1, torch.nn.functional.relu, drop, l2)
    assert type(y.grad_fn).__name__ == "_FusedFFNFnBackward"
    after_fused = torch.cuda.get_rng_state()
    y.backward(gy)
    got = [y.detach(), x.grad.clone()] + [q.grad.clone() for q in (l1.weight, l1.bias, l2.weight, l2.bias)]
    for q in (x, l1.weight, l1.bias, l2.weight, l2.bias):
        q.grad = None
    torch.manual_seed(11)
    y2 = l2(drop(torch.relu(l1(x))))
    assert torch.equal(torch.cuda.get_rng_state(), after_fused), "the node consumed the random stream differently from nn.Dropout"
    y2.backward(gy)
    ref = [y2.detach(), x.grad] + [q.grad for q in (l1.weight, l1.bias, l2.weight, l2.bias)]
    for name, a, b in zip(("out", "grad_x", "grad_w1", "grad_b1", "grad_w2", "grad_b2"), got, ref):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-5, name
    # eval mode / frozen parameters: no dropout, and no weight-gradient launch for a layer that needs none
    drop.eval()
    for q in (l1.weight, l1.bias):
        q.requires_grad_(False)
    x.grad = l1.weight.grad = l1.bias.grad = l2.weight.grad = l2.bias.grad = None
    y3 = fused_ffn(x, l1, torch.nn.functional.relu, drop, l2)
    y3.backward(gy)
    assert l1.weight.grad is None and l2.weight.grad is not None
    assert rel_err(y3.detach().cpu().numpy(), l2(torch.relu(l1(x))).detach().cpu().numpy()) < 2e-5
    # other activations / autocast: the composition of the same modules (no node)
    assert "FusedFFN" not in type(fused_ffn(x, l1, torch.nn.functional.gelu, drop, l2).grad_fn).__name__
