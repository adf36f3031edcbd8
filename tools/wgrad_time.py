#!/usr/bin/env python3
"""Time msda_linear_wgrad_f32 (graph of 10 calls, HIP events) over the module's shapes.
Knobs: MSDA_WGRAD_BIG_M (rows from which the 128x128-tile kernel is used), MSDA_WGRAD_BIG_WGS."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uvhand_amd import _native
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
shapes = [(600, 256, 256), (600, 384, 256), (2400, 256, 256), (6120, 256, 256), (6120, 384, 256), (9600, 256, 256),
          (33440, 256, 256), (33440, 384, 256)]
out = []
with torch.cuda.stream(st):
    for M, N, K in shapes:
        dY, X = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
        fn = lambda: _native.linear_wgrad(dY, X)
        fn(); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                fn()
        for _ in range(3):
            g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(10):
            g.replay()
        b.record(st); b.synchronize()
        us = a.elapsed_time(b) * 1e3 / 100
        out.append("M=%5d N=%3d K=%3d %7.1f us %6.1f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
print("[%s]\n  " % " ".join("%s=%s" % kv for kv in os.environ.items() if kv[0].startswith("MSDA_")) + "\n  ".join(out))
