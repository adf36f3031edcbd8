#!/usr/bin/env python3
"""Host-side cost of one eager call (enqueue only): how fast can Python feed the op."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_inputs
from uvhand_amd import _native
from uvhand_amd.functions import MSDeformAttnFunction

dev = torch.device("cuda", 0)
_, d, dims = make_inputs("cfg2_decoder", 1, dev)
v, l, a = d["value"].requires_grad_(True), d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    el = time.perf_counter() - t0; torch.cuda.synchronize()
    return el / n * 1e6
print("native forward  : %.1f us/call" % t(lambda: _native.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64)))
print("native backward : %.1f us/call" % t(lambda: _native.ms_deform_attn_backward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], d["go"], 64)))
def step():
    v.grad = l.grad = a.grad = None
    MSDeformAttnFunction.apply(v, d["shapes"], d["lsi"], l, a, 64).backward(d["go"])
print("autograd fwd+bwd: %.1f us/step" % t(step, 1000))
with torch.no_grad():
    print("apply (no grad) : %.1f us/call" % t(lambda: MSDeformAttnFunction.apply(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64)))
print("torch.empty x3  : %.1f us" % t(lambda: (torch.empty_like(d["value"]), torch.empty_like(d["loc"]), torch.empty_like(d["attn"]))))
