#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  It imports
the reference's own pure-PyTorch implementation of the hot path
(models/ops/functions/ms_deform_attn_func.py:42-62, ``ms_deform_attn_core_pytorch``)
and, for the module-level cases, the reference module
(models/ops/modules/ms_deform_attn.py:30-140) with its CUDA autograd function
replaced by that same fallback (the CUDA extension cannot be built here: no
nvcc / CUDA device).  Forward outputs come from the reference function; the
three gradients come from torch.autograd through it.  Nothing of the reference
is copied: the .npz files hold inputs and expected outputs only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Cases:
  testpy_double / testpy_float     models/ops/test.py:31-60, seed 3, exact draw order
  testpy_grad_D{30,...,3096}       models/ops/test.py:63-86 inputs, analytic fp64 grads
  cfg1                             BASELINE config 1 (N=1, 32x32+16x16, Lq=100, M=8, D=32, P=4)
  cfg2_sub                         cfg-2 geometry (48/24/12/6, M=8, D=32, P=4), N=1, Lq=16
  oob                              locations ~ U(-0.5, 1.5): zero padding, partial taps
  edges                            pixel centres and the -1 / W-1 / W boundary values
  chunk                            N=4 (im2col_step=2 < N in the tests)
  module_state                     the perturbed state_dict the two module cases use
  module_2d / module_42d           MSDeformAttn.forward (+ all grads) with module_state
  module_init                      state_dict right after construction under manual_seed(0)
"""
import os
import sys
import types

import numpy as np
import torch

REF_OPS = "/root/reference/models/ops"
HERE = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
sys.path.insert(0, os.path.dirname(REF_OPS))          # -> `import ops...` (namespace package)
from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch as ref_core  # noqa: E402
import ops.modules.ms_deform_attn as ref_mod                                            # noqa: E402


def level_start(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def run_ref(value, shapes, loc, attn, grad_out, dtype):
    v = value.to(dtype).clone().requires_grad_(True)
    l = loc.to(dtype).clone().requires_grad_(True)
    a = attn.to(dtype).clone().requires_grad_(True)
    out = ref_core(v, shapes, l, a)
    out.backward(grad_out.to(dtype))
    return out.detach(), v.grad, l.grad, a.grad


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-22s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def rand_inputs(N, S, M, D, Lq, L, P):
    """Exactly the construction of models/ops/test.py:33-36 (CPU generator)."""
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, loc, attn


def op_case(name, value, shapes, loc, attn, grad_out, store32=False, **extra):
    """Store fp32 inputs (they originate as fp32, so this is lossless) and the
    reference's fp64 results; with store32 also its fp32 results."""
    o, gv, gl, ga = run_ref(value, shapes, loc, attn, grad_out, torch.float64)
    arrs = dict(value=value, shapes=shapes, level_start=level_start(shapes), loc=loc, attn=attn,
                grad_out=grad_out, out=o, grad_value=gv, grad_loc=gl, grad_attn=ga, **extra)
    if store32:
        o32, gv32, gl32, ga32 = run_ref(value, shapes, loc, attn, grad_out, torch.float32)
        arrs.update(out_f32=o32, grad_value_f32=gv32, grad_loc_f32=gl32, grad_attn_f32=ga32)
    save(name, **arrs)


def main():
    # ---- models/ops/test.py, exact RNG stream -------------------------------------------
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value, loc, attn = rand_inputs(N, S, M, D, Lq, L, P)                 # test.py:33-36
    g = torch.Generator().manual_seed(1234)
    go = torch.rand(N, Lq, M * D, generator=g)
    op_case("testpy_double", value, shapes, loc, attn, go)
    value, loc, attn = rand_inputs(N, S, M, D, Lq, L, P)                 # test.py:49-52
    op_case("testpy_float", value, shapes, loc, attn, go, store32=True)
    for ch in [30, 32, 64, 71, 1025, 2048, 3096]:                        # test.py:85
        value, loc, attn = rand_inputs(N, S, M, ch, Lq, L, P)            # test.py:65-68
        go = torch.rand(N, Lq, M * ch, generator=g)
        op_case("testpy_grad_D%d" % ch, value, shapes, loc, attn, go)

    # ---- BASELINE config 1 ---------------------------------------------------------------
    g = torch.Generator().manual_seed(0)

    def gen(N, shapes, M, D, Lq, P, lo=0.0, hi=1.0):
        L = shapes.shape[0]
        S = int(shapes.prod(1).sum())
        value = torch.rand(N, S, M, D, generator=g) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * (hi - lo) + lo
        attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        go = torch.rand(N, Lq, M * D, generator=g)
        return value, loc, attn, go

    shapes = torch.as_tensor([(32, 32), (16, 16)], dtype=torch.long)
    value, loc, attn, go = gen(1, shapes, 8, 32, 100, 4)
    op_case("cfg1", value, shapes, loc, attn, go)

    # ---- cfg-2 geometry, subsampled in N and Lq; locations spill outside the maps ---------
    shapes = torch.as_tensor([(48, 48), (24, 24), (12, 12), (6, 6)], dtype=torch.long)
    value, loc, attn, go = gen(1, shapes, 8, 32, 16, 4, lo=-0.25, hi=1.25)
    op_case("cfg2_sub", value, shapes, loc, attn, go)

    # ---- out-of-range locations, odd sizes --------------------------------------------------
    shapes = torch.as_tensor([(5, 7), (3, 4), (1, 2)], dtype=torch.long)
    value, loc, attn, go = gen(2, shapes, 3, 8, 9, 3, lo=-0.5, hi=1.5)
    op_case("oob", value, shapes, loc, attn, go, store32=True)

    # ---- boundary / integer-centre locations ---------------------------------------------
    # pixel coordinate k (so that loc*W-0.5 == k) for k in a list that brackets every guard of
    # ms_deform_im2col_cuda.cuh:56-78 and :288.  k == -1 exactly is where the reference's CUDA
    # kernel (skips the point, :288) and its grid_sample fallback (keeps a zero-weight in-range
    # tap, so grad_loc != 0) disagree on grad_loc; `exact_m1` marks those points.
    shapes = torch.as_tensor([(4, 6), (2, 3)], dtype=torch.long)
    ks = [-1.5, -1.0, -0.75, -0.5, 0.0, 0.5, 1.0, 2.0]
    L, P, M, D, N = 2, 4, 2, 4, 1
    pts = []
    for l in range(L):
        H, W = [int(x) for x in shapes[l]]
        kx = ks + [W - 2.0, W - 1.5, W - 1.0, W - 0.5, W - 0.25, float(W)]
        ky = ks + [H - 2.0, H - 1.5, H - 1.0, H - 0.5, H - 0.25, float(H)]
        pts.append([((x + 0.5) / W, (y + 0.5) / H) for x in kx for y in ky])
    Lq = (len(pts[0]) + P - 1) // P
    loc = torch.full((N, Lq, M, L, P, 2), 0.5)
    for l in range(L):
        for i, (x, y) in enumerate(pts[l]):
            q, p = divmod(i, P)
            loc[0, q, :, l, p, 0] = x
            loc[0, q, :, l, p, 1] = y
    loc[:, :, 1] = loc[:, :, 1].flip(1)       # second head walks the list backwards
    S = int(shapes.prod(1).sum())
    value = torch.rand(N, S, M, D, generator=g) + 0.5
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    go = torch.rand(N, Lq, M * D, generator=g)
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(torch.float32)[None, None, None, :, None, :]
    pix32 = loc * wh - 0.5
    pix64 = loc.double() * wh.double() - 0.5
    exact_m1 = ((pix32 == -1) | (pix64 == -1)).any(-1)
    op_case("edges", value, shapes, loc, attn, go, store32=True, exact_m1=exact_m1)

    # ---- N > im2col_step -------------------------------------------------------------------
    shapes = torch.as_tensor([(8, 8), (4, 4)], dtype=torch.long)
    value, loc, attn, go = gen(4, shapes, 4, 16, 12, 2, lo=-0.1, hi=1.1)
    op_case("chunk", value, shapes, loc, attn, go, store32=True)

    # ---- module level ------------------------------------------------------------------------
    class _FallbackFn:
        """Stands in for the CUDA autograd Function inside the reference module."""
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, im2col_step):
            return ref_core(value, shapes, loc, attn)

    ref_mod.MSDeformAttnFunction = _FallbackFn

    torch.manual_seed(0)
    mod = ref_mod.MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)
    save("module_init", **{k: v for k, v in mod.state_dict().items()})

    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for p in mod.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)
    save("module_state", **{k: v.clone() for k, v in mod.state_dict().items()})
    shapes = torch.as_tensor([(8, 8), (4, 4), (2, 2), (1, 1)], dtype=torch.long)
    lsi = level_start(shapes)
    S = int(shapes.prod(1).sum())
    N, Lq = 2, 10
    for name, refdim in (("module_2d", 2), ("module_42d", 42)):
        query = torch.randn(N, Lq, 256, generator=g, requires_grad=True)
        src = torch.randn(N, S, 256, generator=g, requires_grad=True)
        refp = (torch.rand(N, Lq, 4, refdim, generator=g) * 1.4 - 0.2).requires_grad_(True)
        mask = torch.zeros(N, S, dtype=torch.bool)
        mask[1, -7:] = True
        gout = torch.randn(N, Lq, 256, generator=g)
        mod.zero_grad()
        out = mod(query, refp, src, shapes, lsi, mask)
        out.backward(gout)
        arrs = {"pgrad." + k: p.grad for k, p in mod.named_parameters()}
        save(name, query=query, src=src, refp=refp, mask=mask, shapes=shapes, level_start=lsi,
             gout=gout, out=out, grad_query=query.grad, grad_src=src.grad, grad_refp=refp.grad, **arrs)


if __name__ == "__main__":
    main()
