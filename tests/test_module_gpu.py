"""Module-level parity on the GPU: uvhand_amd.modules.MSDeformAttn against golden vectors captured from
the reference module (models/ops/modules/ms_deform_attn.py:80-140) with a fixed state_dict."""
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


def test_module_bf16_storage_option():
    mod = _module()
    z = load_golden("module_2d")
    args = [torch.from_numpy(z[k]).cuda() for k in ("query", "refp", "src", "shapes", "level_start")]
    ref = mod(*args)
    mod.bf16_storage = True
    q = args[0].clone().requires_grad_(True)
    out = mod(q, *args[1:])
    out.sum().backward()
    assert out.dtype == torch.float32 and torch.isfinite(q.grad).all()
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-2
