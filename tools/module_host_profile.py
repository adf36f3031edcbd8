#!/usr/bin/env python3
"""Where the eager module step's HOST time goes (cfg2_decoder by default): torch.profiler, CPU activities only, self time per op
over 200 forward+backward steps of one MSDeformAttn module (C++ node)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from bench import WORKLOADS
from uvhand_amd.modules import MSDeformAttn

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_decoder"
N, shapes, M, D, Lq, P = WORKLOADS[name]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mod = MSDeformAttn(M * D, len(shapes), M, P).to(dev)
sh = torch.tensor(shapes, dtype=torch.long, device=dev)
lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
S = int(sh.prod(1).sum())
q = torch.randn(N, Lq, M * D, device=dev, requires_grad=True)
ref = torch.rand(N, Lq, len(shapes), 2, device=dev)
x = torch.randn(N, S, M * D, device=dev, requires_grad=True)
go = torch.randn(N, Lq, M * D, device=dev)


def step():
    for p in mod.parameters():
        p.grad = None
    q.grad = x.grad = None
    mod(q, ref, x, sh, lsi, None).backward(go)


for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("eager wall %.1f us per step" % (1e6 * (time.perf_counter() - t0) / 200))
t0 = time.perf_counter()
for _ in range(200):
    out = mod(q, ref, x, sh, lsi, None)
t1 = time.perf_counter()
print("forward enqueue %.1f us" % (1e6 * (t1 - t0) / 200))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(200):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=60))
