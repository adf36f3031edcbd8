#!/usr/bin/env python3
"""Transformer input assembly (f3): the native flatten kernel against the reference's composition, fwd + bwd, training shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uvhand_amd import _native
from uvhand_amd.utils.transformer_inputs import _FlattenLevelsFn

def bench(fn, iters=50):
    """GPU time per call: the call is captured into a HIP graph (the eager loop is host-bound either way)."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn()
        for _ in range(3): g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(iters): g.replay()
        b.record(st); b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

for N, shapes in ((32, [(28, 28), (14, 14), (7, 7), (4, 4)]), (2, [(48, 48), (24, 24), (12, 12), (6, 6)])):
    C = 256
    srcs = [torch.randn(N, C, h, w, device="cuda", requires_grad=True) for h, w in shapes]
    poss = [torch.randn(N, C, h, w, device="cuda", requires_grad=True) for h, w in shapes]
    masks = [torch.zeros(N, h, w, dtype=torch.bool, device="cuda") for h, w in shapes]
    embed = torch.randn(len(shapes), C, device="cuda", requires_grad=True)
    S = sum(h * w for h, w in shapes)
    g1, g2 = torch.randn(N, S, C, device="cuda"), torch.randn(N, S, C, device="cuda")
    def native():
        s, p = _FlattenLevelsFn.apply(embed, *srcs, *poss)      # (the shape tensors are host->device copies: not capturable)
        torch.autograd.grad([s, p], srcs + poss + [embed], [g1, g2])
    def stock():
        s = torch.cat([t.flatten(2).transpose(1, 2) for t in srcs], 1)
        p = torch.cat([t.flatten(2).transpose(1, 2) + embed[l].view(1, 1, -1) for l, t in enumerate(poss)], 1)
        torch.autograd.grad([s, p], srcs + poss + [embed], [g1, g2])
    tn, ts = bench(native), bench(stock)
    nbytes = 16 * N * S * C * 2                      # two tensors, read + write, forward and backward
    print("N=%d S=%d C=%d: native %.1f us (%.2f TB/s of algorithmic bytes), reference composition %.1f us" % (N, S, C, tn, nbytes / tn / 1e6, ts))
