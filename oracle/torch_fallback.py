"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Own restatement, on stock PyTorch CPU ops, of the reference's pure-PyTorch
fallback ``ms_deform_attn_core_pytorch``
(/root/reference/models/ops/functions/ms_deform_attn_func.py:42-62): per level a
``grid_sample(bilinear, zeros, align_corners=False)`` of the level's feature map
at ``2*loc-1``, weighted by the attention weights and summed over levels x
points; backward by autograd.  This is what BASELINE.md names as the CPU
comparator ("the repo's pure-PyTorch fallback timed on host cores"); the
reference file itself cannot travel to the GPU box, so bench.py times this port
(``cpu_baseline.kind == "port"``).  tests/test_oracle.py pins it to the golden
vectors generated from the reference's function.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.
"""
import torch
import torch.nn.functional as F


def msda_torch_fallback(value, spatial_shapes, sampling_locations, attention_weights):
    """value[N,S,M,D], spatial_shapes[L,2] (H,W), sampling_locations[N,Lq,M,L,P,2] (x,y in
    normalised coords), attention_weights[N,Lq,M,L,P] -> out[N,Lq,M*D]."""
    n, s, m, d = value.shape
    lq, n_levels, n_points = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    hw = [(int(h), int(w)) for h, w in spatial_shapes.tolist()] if torch.is_tensor(spatial_shapes) \
        else [(int(h), int(w)) for h, w in spatial_shapes]
    # heads become the batch of grid_sample: [N*M, D, S]
    maps = value.permute(0, 2, 3, 1).reshape(n * m, d, s)
    grid = (sampling_locations * 2 - 1).permute(0, 2, 1, 3, 4, 5).reshape(n * m, lq, n_levels, n_points, 2)
    weights = attention_weights.permute(0, 2, 1, 3, 4).reshape(n * m, 1, lq, n_levels, n_points)
    acc = None
    start = 0
    for lvl, (h, w) in enumerate(hw):
        fmap = maps[:, :, start:start + h * w].reshape(n * m, d, h, w)
        start += h * w
        sampled = F.grid_sample(fmap, grid[:, :, lvl], mode="bilinear", padding_mode="zeros",
                                align_corners=False)                 # [N*M, D, Lq, P]
        part = (sampled * weights[:, :, :, lvl]).sum(-1)             # [N*M, D, Lq]
        acc = part if acc is None else acc + part
    return acc.reshape(n, m * d, lq).transpose(1, 2).contiguous()


def fwd_bwd(value, spatial_shapes, sampling_locations, attention_weights, grad_output):
    """One forward + backward of the fallback; returns (out, grad_value, grad_loc, grad_attn)."""
    v = value.detach().clone().requires_grad_(True)
    l = sampling_locations.detach().clone().requires_grad_(True)
    a = attention_weights.detach().clone().requires_grad_(True)
    out = msda_torch_fallback(v, spatial_shapes, l, a)
    out.backward(grad_output)
    return out.detach(), v.grad, l.grad, a.grad
