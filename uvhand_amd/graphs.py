"""``uvhand_amd.graphed`` — the module (or any callable built on this package's ops) as a pair of HIP graphs.

At decoder sizes one ``MSDeformAttn`` forward + backward is ~110 us of GPU work behind 190-380 us of host work
(autograd's node hand-off, ~20 launches, a dozen allocations — DESIGN.md section 5): the eager training loop of the
reference (``engine.py:590-648``) is host-bound there.  ``torch.cuda.make_graphed_callables`` removes the host from the
picture — forward and backward are each captured once and replayed — and the ops of this package are capture-safe by
construction: they launch on the current stream, never synchronise, allocate only through torch's caching allocator (which
serves a capture from a private pool) and read ``spatial_shapes`` / ``level_start_index`` on the device.  The one host-side
check of the module — ``sum(H_l * W_l) == Len_in``, a device-to-host read the reference performs on every call
(``models/ops/modules/ms_deform_attn.py:93``) — is done here, once, before the capture.

    attn = uvhand_amd.graphed(attn, (query, reference_points, src, spatial_shapes, level_start_index))
    out = attn(query, reference_points, src, spatial_shapes, level_start_index)      # same signature, same autograd

Constraints are ``torch.cuda.make_graphed_callables``'s: fixed shapes, positional tensor arguments, no data-dependent
control flow, not inside ``torch.autocast`` capture without ``cache_enabled=False``.  Dropout inside a graphed callable draws
from the graph-safe Philox offsets torch registers for the capture, so masks still change from replay to replay.

One pitfall of this PyTorch-ROCm build, found while testing this wrapper (and reproduced with a bare ``nn.Linear``): calling
``make_graphed_callables`` while the output of an earlier EAGER forward + backward through the same parameters is still alive
crashes inside the capture of the backward graph.  ``graphed`` therefore collects garbage and synchronises before capturing;
drop references to earlier outputs (``del out``) before calling it.
"""
import gc

import torch

from .modules.ms_deform_attn import MSDeformAttn, _check_shapes_sum


def _precheck(module, sample_args):
    """What the module would read back from the device on its first call: done before the capture (nothing may synchronise
    during one).  Works for a bare MSDeformAttn and for containers of them when the pyramid tensors are among the sample arguments."""
    int64s = [a for a in sample_args if torch.is_tensor(a) and a.dtype == torch.int64 and a.dim() == 2 and a.shape[-1] == 2]
    feats = [a for a in sample_args if torch.is_tensor(a) and a.is_floating_point() and a.dim() == 3]
    for shapes in int64s:
        total = int((shapes[:, 0] * shapes[:, 1]).sum())
        for f in feats:
            if f.shape[1] == total:
                _check_shapes_sum(shapes, total)


def graphed(callables, sample_args, num_warmup_iters=3, allow_unused_input=True, pool=None):
    """``torch.cuda.make_graphed_callables`` for modules / functions built on this package (one callable and one tuple of
    sample arguments, or tuples of each): returns callable(s) with the same signature whose forward and backward replay HIP
    graphs.  Sample arguments must have the shapes, dtypes and ``requires_grad`` flags of the real ones; integer tensors
    (``spatial_shapes``, ``level_start_index``) and masks are passed as they are.  Results equal the eager call's: the same
    kernels run, in the same order, on the same stream."""
    single = not isinstance(callables, (tuple, list))
    cs = (callables,) if single else tuple(callables)
    args = (tuple(sample_args),) if single else tuple(tuple(a) for a in sample_args)
    if len(cs) != len(args):
        raise ValueError("graphed: one tuple of sample arguments per callable")
    for c, a in zip(cs, args):
        if not all(torch.is_tensor(t) for t in a):
            raise TypeError("graphed: every sample argument must be a tensor (torch.cuda.make_graphed_callables' rule)")
        if not any(t.is_cuda for t in a):
            raise RuntimeError("graphed: the sample arguments are not on a GPU")
        if isinstance(c, torch.nn.Module):
            _precheck(c, a)
    gc.collect()                                             # (see the module docstring: no stale eager graphs across the capture)
    torch.cuda.synchronize()
    out = torch.cuda.make_graphed_callables(cs if not single else cs[0], args if not single else args[0],
                                            num_warmup_iters=num_warmup_iters, allow_unused_input=allow_unused_input, pool=pool)
    return out


__all__ = ["graphed", "MSDeformAttn"]
