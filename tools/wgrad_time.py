#!/usr/bin/env python3
"""GPU time of the fp32 weight gradient dW = dY^T X (+ bias) at the layers' shapes: msda_linear_wgrad_f32 (both stages) against
torch's mm + sum (hipBLASLt).  Graph of 10 calls, HIP events.  Any MSDA_* knob selects the tuning library
(MSDA_WGRAD_BIG_ROWS=0: without the 128-tile kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uvhand_amd import _native
if any(k.startswith("MSDA_") for k in os.environ):
    _native.LIB_PATH = os.path.join(ROOT, "uvhand_amd", "libmsda_hip_tuning.so")
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
shapes = [(6120, 256, 256), (9600, 256, 256), (9600, 1024, 256), (33440, 256, 256), (33440, 384, 256), (33440, 1024, 256), (33440, 256, 1024)]


def gpu_us(fn):
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
    return a.elapsed_time(b) * 1e3 / 100


print("rows  out   in | ours us (TFLOP/s) | torch mm + sum us (TFLOP/s)   [%s]" % " ".join("%s=%s" % kv for kv in os.environ.items() if kv[0].startswith("MSDA_")))
with torch.cuda.stream(st):
    for rows, out_f, in_f in shapes:
        x, gy = torch.randn(rows, in_f, device=dev), torch.randn(rows, out_f, device=dev)
        fl = 2.0 * rows * out_f * in_f / 1e6
        ours = lambda: _native.linear_wgrad(gy, x)
        ref = lambda: (gy.t().mm(x), gy.sum(0))
        a, b = min(gpu_us(ours), gpu_us(ours)), min(gpu_us(ref), gpu_us(ref))
        print("%5d %4d %4d | %7.1f (%5.1f) | %7.1f (%5.1f)" % (rows, out_f, in_f, a, fl / a, b, fl / b), flush=True)
