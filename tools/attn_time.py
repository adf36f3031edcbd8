#!/usr/bin/env python3
"""GPU time of the decoder self-attention core at the training shape (300 queries, 32 frames x 8 heads of 32, dropout 0.1):
msda_attn32_forward_f32 / msda_attn32_backward_f32 against torch's scaled_dot_product_attention (forward, and forward+backward).
Graph of 10 calls, HIP events."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from uvhand_amd import _native
if any(k.startswith("MSDA_") for k in os.environ):
    _native.LIB_PATH = os.path.join(ROOT, "uvhand_amd", "libmsda_hip_tuning.so")
if os.environ.get("ATTN_LIB"):                               # a library built with other compile-time constants
    _native.LIB_PATH = os.path.join(ROOT, os.environ["ATTN_LIB"])
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
L, N, H = int(os.environ.get("ATTN_L", 300)), int(os.environ.get("ATTN_N", 32)), 8
p = float(os.environ.get("ATTN_P", 0.1))


def gpu_us(fn):
    fn(); st.synchronize()
    if os.environ.get("ATTN_EAGER"):                      # plain launches (counter passes: tools/attn_pmc.sh)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(20):
            fn()
        b.record(st); b.synchronize()
        return a.elapsed_time(b) * 1e3 / 20
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


with torch.cuda.stream(st):
    qk, v, go = torch.randn(L, N, 512, device=dev), torch.randn(L, N, 256, device=dev), torch.randn(L, N, 256, device=dev)
    q, k = qk[..., :256], qk[..., 256:]
    seed = torch.tensor([12345], dtype=torch.int64, device=dev)
    scale = 1 / math.sqrt(32)
    out, lse = _native.attn32_forward(q, k, v, H, scale, p, seed)
    gqk = torch.empty_like(qk)
    gv = torch.empty_like(v)
    fwd = lambda: _native.attn32_forward(q, k, v, H, scale, p, seed)
    bwd = lambda: _native.attn32_backward(q, k, v, out, lse, go, H, scale, p, seed, grad_q=gqk[..., :256], grad_k=gqk[..., 256:], grad_v=gv)
    tf, tb = min(gpu_us(fwd), gpu_us(fwd)), min(gpu_us(bwd), gpu_us(bwd))
    flops = 4.0 * N * H * L * L * 32
    print("ours : forward %6.1f us (%5.1f TFLOP/s)  backward %6.1f us (%5.1f TFLOP/s)" % (tf, flops / tf / 1e6, tb, 2.5 * flops / tb / 1e6))
    sp = lambda t: t.reshape(L, N * H, 32).transpose(0, 1)
    q4, k4, v4 = (sp(t.contiguous()).reshape(N, H, L, 32).detach().requires_grad_(True) for t in (q, k, v))
    g4 = torch.randn(N, H, L, 32, device=dev)
    sd = lambda: F.scaled_dot_product_attention(q4, k4, v4, dropout_p=p)
    def sd_fb():
        q4.grad = k4.grad = v4.grad = None
        sd().backward(g4)
    with torch.no_grad():
        tsf = min(gpu_us(sd), gpu_us(sd))
    tsb = min(gpu_us(sd_fb), gpu_us(sd_fb))
    print("torch: forward %6.1f us                  forward+backward %6.1f us   (ours %6.1f)" % (tsf, tsb, tf + tb))
