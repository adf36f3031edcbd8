#!/usr/bin/env python3
"""What stock PyTorch-ROCm (hipBLASLt / rocBLAS) achieves on the fp32 GEMMs that bracket the op."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uvhand_amd import _native
dev = torch.device("cuda", 0)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for M in (600, 6120, 33440):
    for N in (128, 256):
        K = 256
        X, W, dY, bias = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev), torch.randn(N, device=dev)
        fl = 2.0 * M * N * K
        r = [("fwd  Y=X@W^T+b", lambda: torch.addmm(bias, X, W.t())), ("dgrad dX=dY@W", lambda: dY @ W), ("wgrad dW=dY^T@X", lambda: dY.t() @ X),
             ("bias grad", lambda: dY.sum(0)), ("OURS wgrad+bias (msda_linear_wgrad_f32)", lambda: _native.linear_wgrad(dY, X))]
        print("M=%5d N=%3d K=%3d: " % (M, N, K) + "  ".join("%s %.1f us (%.1f TF)" % (n, u, fl / u / 1e6) for n, u in ((n, t(f)) for n, f in r)))
