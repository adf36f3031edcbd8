#!/usr/bin/env python3
"""Time msda_linear_wgrad_{f32, masked_bf16} (graph of 10 calls, HIP events) over the module's and the FFN's shapes.
    python tools/wgrad_time.py [f32|bf16]      (MSDA_* knobs: diagnostic library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uvhand_amd import _native
if any(k.startswith("MSDA_") for k in os.environ):
    _native.LIB_PATH = os.path.join(ROOT, "uvhand_amd", "libmsda_hip_tuning.so")
dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
shapes = [(600, 256, 256), (600, 384, 256), (2400, 256, 256), (6120, 256, 256), (6120, 384, 256), (9600, 256, 256),
          (33440, 256, 256), (33440, 384, 256), (33440, 1024, 256), (33440, 256, 1024)]
out = []
with torch.cuda.stream(st):
    for M, N, K in shapes:
        dY, X = torch.randn(M, N, device=dev).to(dtype), torch.randn(M, K, device=dev).to(dtype)
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
print("%s [%s]\n  " % (str(dtype), " ".join("%s=%s" % kv for kv in os.environ.items() if kv[0].startswith("MSDA_"))) + "\n  ".join(out))
