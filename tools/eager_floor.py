#!/usr/bin/env python3
"""Where the eager step's host time goes (headline shape): wall time per step, back-to-back calls, for
  a. an autograd Function that only allocates its outputs (the PyTorch floor for this signature: one node, three gradients),
  b. the shipped op (C++ node in _msda_torch.so),
  c. the shipped op's forward alone under no_grad,
  d. the ctypes path without autograd (forward with table + backward),
each as the median of five blocks of 200 steps."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_inputs
from uvhand_amd import _native
from uvhand_amd.functions import MSDeformAttnFunction


class Floor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, shapes, lsi, loc, attn, step):
        ctx.save_for_backward(value, shapes, lsi, loc, attn)
        return value.new_empty(value.shape[0], loc.shape[1], value.shape[2] * value.shape[3])

    @staticmethod
    def backward(ctx, go):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        return torch.empty_like(value), None, None, torch.empty_like(loc), torch.empty_like(attn), None


def block(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


def main():
    dev = torch.device("cuda", 0)
    _, d, _ = make_inputs(sys.argv[1] if len(sys.argv) > 1 else "cfg2_decoder", 1000, dev, "uniform")
    value, loc, attn = d["value"].requires_grad_(True), d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
    sh, lsi, go = d["shapes"], d["lsi"], d["go"]

    def step_of(apply):
        def step():
            value.grad = loc.grad = attn.grad = None
            apply(value, sh, lsi, loc, attn, 64).backward(go)
        return step

    def fwd_only():
        with torch.no_grad():
            MSDeformAttnFunction.apply(value, sh, lsi, loc, attn, 64)

    vd, ld, ad = value.detach(), loc.detach(), attn.detach()

    def raw():
        out, table = _native.ms_deform_attn_forward(vd, sh, lsi, ld, ad, 64, with_table=True)
        _native.ms_deform_attn_backward(vd, sh, lsi, ld, ad, go, 64, table=table)

    rows = {}
    for name, fn in (("floor_function", step_of(Floor.apply)), ("shipped_op", step_of(MSDeformAttnFunction.apply)),
                     ("shipped_forward_no_grad", fwd_only), ("ctypes_no_autograd", raw)):
        for _ in range(20):
            fn()
        rows[name + "_us"] = round(sorted(block(fn, 200) for _ in range(5))[2], 2)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
