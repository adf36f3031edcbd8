#!/usr/bin/env python3
"""Time msda_linear_forward_f32 / msda_linear_dgrad_f32 against torch's addmm / mm (hipBLASLt) at the module's shapes: GPU time
(graph of 10 calls, HIP events) and host time per eager call.
    python tools/gemm_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uvhand_amd import _native
if any(k.startswith("MSDA_") for k in os.environ):
    _native.LIB_PATH = os.path.join(ROOT, "uvhand_amd", "libmsda_hip_tuning.so")
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
shapes = [(600, 256, 256), (600, 384, 256), (6120, 256, 256), (9600, 256, 256), (33440, 256, 256), (33440, 384, 256),
          (33440, 1024, 256), (33440, 256, 1024)]
if os.environ.get("GEMM_SHAPES") == "small":
    shapes = [(300, 256, 256), (1200, 256, 256), (1200, 384, 256), (2400, 256, 256), (2400, 384, 256), (3600, 256, 256), (4800, 256, 256)]


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


def host_us(fn, n=200):
    for _ in range(10):
        fn()
    st.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t; st.synchronize()
    return 1e6 * dt / n


print("rows  out   in | forward: ours us (TF) host us | torch us (TF) host us || dgrad: ours us (TF) | torch us (TF)")
with torch.cuda.stream(st):
    for rows, out_f, in_f in shapes:
        x, w, b = torch.randn(rows, in_f, device=dev), torch.randn(out_f, in_f, device=dev) * 0.1, torch.randn(out_f, device=dev)
        gy = torch.randn(rows, out_f, device=dev)
        wt = w.t()
        fl = 2.0 * rows * out_f * in_f / 1e6
        f_ours, f_t = (lambda: _native.linear_forward(x, w, b)), (lambda: torch.addmm(b, x, wt))
        d_ours, d_t = (lambda: _native.linear_dgrad(gy, w)), (lambda: torch.mm(gy, w))
        a, bb, c, d = gpu_us(f_ours), gpu_us(f_t), gpu_us(d_ours), gpu_us(d_t)
        print("%5d %4d %4d | %7.1f (%5.1f) %5.1f | %7.1f (%5.1f) %5.1f || %7.1f (%5.1f) | %7.1f (%5.1f)" % (
            rows, out_f, in_f, a, fl / a, host_us(f_ours), bb, fl / bb, host_us(f_t), c, fl / c, d, fl / d))
