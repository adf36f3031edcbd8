#!/usr/bin/env python3
"""Diagnostic: host-side cost of one eager MSDeformAttn module forward + backward at the headline decoder shape (torch.profiler,
CPU ops by self time) — where the 300+ us of wall time per step go while the kernels need ~120 us."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS
from uvhand_amd.modules import MSDeformAttn
wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2_decoder"
N, shapes, M, D, Lq, P = WORKLOADS[wl]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mod = MSDeformAttn(M * D, len(shapes), M, P).to(dev)
mod.cpp_node = "--no-cpp-node" not in sys.argv
amp = "--amp" in sys.argv                      # autocast(bfloat16) + bf16 rows, as bench.py's `modules` rows with amp
mod.bf16_storage = amp
sh = torch.tensor(shapes, dtype=torch.long, device=dev)
lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
S = int(sh.prod(1).sum())
q = torch.randn(N, Lq, M * D, device=dev, requires_grad=True)
src = torch.randn(N, S, M * D, device=dev, requires_grad=True)
ref = torch.rand(N, Lq, len(shapes), 2, device=dev, requires_grad=True)
go = torch.randn(N, Lq, M * D, device=dev)


def step():
    mod.zero_grad(set_to_none=True)
    q.grad = src.grad = ref.grad = None
    if amp:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = mod(q, ref, src, sh, lsi)
        out.backward(go.to(out.dtype))
    else:
        mod(q, ref, src, sh, lsi).backward(go)


for _ in range(20):
    step()
gc.collect(); gc.freeze()
torch.cuda.synchronize()
for label, n in (("first", 200), ("second", 200)):
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%s block: host enqueue %.1f us / step, with the final wait %.1f us / step" % (label, 1e6 * (t1 - t0) / n, 1e6 * (time.perf_counter() - t0) / n))
# forward and backward separately (host time only)
tf = tb = 0.0
for _ in range(100):
    mod.zero_grad(set_to_none=True); q.grad = src.grad = ref.grad = None
    torch.cuda.synchronize(); a = time.perf_counter()
    if amp:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = mod(q, ref, src, sh, lsi)
    else:
        out = mod(q, ref, src, sh, lsi)
    b = time.perf_counter()
    out.backward(go.to(out.dtype))
    c = time.perf_counter()
    tf += b - a; tb += c - b
print("host: forward %.1f us, backward %.1f us (GPU idle at the start of each)" % (1e4 * tf, 1e4 * tb))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=70))
