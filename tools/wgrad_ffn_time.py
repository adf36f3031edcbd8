#!/usr/bin/env python3
"""Time msda_linear_wgrad_f32 at the encoder's shapes incl. the FFN (graph of 10 calls).  MSDA_WGRAD_WGS (diagnostic library)
= workgroups aimed for."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uvhand_amd import _native
if any(k.startswith("MSDA_") for k in os.environ):
    _native.LIB_PATH = os.path.join(ROOT, "uvhand_amd", "libmsda_hip_tuning.so")
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
out = []
with torch.cuda.stream(st):
    for M, N, K in [(33440, 256, 256), (33440, 384, 256), (33440, 1024, 256), (33440, 256, 1024), (9600, 1024, 256), (9600, 256, 256)]:
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
        out.append("M=%5d N=%4d K=%4d %7.1f us %6.1f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
print("[%s]\n  " % " ".join("%s=%s" % kv for kv in os.environ.items() if kv[0].startswith("MSDA_")) + "\n  ".join(out))
