#!/usr/bin/env python3
"""One MSDeformAttn MODULE forward+backward loop (the op plus its four nn.Linear, softmax and location
arithmetic) for profiling:  rocprofv3 --kernel-trace --stats -- python3 tools/module_step.py cfg4_encoder"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS
from uvhand_amd.modules import MSDeformAttn

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_encoder"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
amp = os.environ.get("MODULE_AMP", "")            # "bf16": autocast + bf16 storage in the op
fused = os.environ.get("MODULE_FUSED_PROLOGUE", "1") != "0"
merged = os.environ.get("MODULE_MERGED_PROJECTION", "1") != "0"
N, shapes, M, D, Lq, P = WORKLOADS[name]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mod = MSDeformAttn(M * D, len(shapes), M, P).to(dev)
mod.fused_prologue = fused
mod.merged_projection = merged
mod.cpp_node = os.environ.get("MODULE_CPP_NODE", "1") != "0"        # 0: the Python composition of the same kernels
with torch.no_grad():
    for p in mod.parameters():
        p.add_(torch.randn_like(p) * 0.02)
sh = torch.tensor(shapes, dtype=torch.long, device=dev)
lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
S = int(sh.prod(1).sum())
q = torch.randn(N, Lq, M * D, device=dev, requires_grad=True)
ref_requires_grad = True
src = torch.randn(N, S, M * D, device=dev, requires_grad=True)
ref = torch.rand(N, Lq, len(shapes), 2, device=dev, requires_grad=True)
go = torch.randn(N, Lq, M * D, device=dev)
# MODULE_MASK=1: a padding mask with ~5 % of the pixels flagged (MSDA_MASKED_ROWS=0 -> reference's masked_fill)
mask = (torch.rand(N, S, device=dev) < 0.05) if os.environ.get("MODULE_MASK", "") == "1" else None
def step():
    mod.zero_grad(set_to_none=True); q.grad = src.grad = ref.grad = None
    if amp == "bf16":
        mod.bf16_storage = True
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = mod(q, ref, src, sh, lsi, mask)
        out.backward(go.to(out.dtype))
    else:
        out = mod(q, ref, src, sh, lsi, mask)
        out.backward(go)

use_graph = os.environ.get("MODULE_GRAPH", "") == "1"              # replay a HIP graph: GPU-bound time
st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    for _ in range(3):
        step()
    st.synchronize()
    run = step
    if use_graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            step()
        run = g.replay
    for _ in range(3):
        run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    import time
    st.synchronize(); t0 = time.perf_counter()
    e0.record(st)
    for _ in range(iters):
        run()
    t_enq = time.perf_counter() - t0
    e1.record(st); e1.synchronize()
print("%s module fwd+bwd%s%s%s: %.1f us per step (host enqueue %.1f us per step)" % (
      name, " (autocast bf16 + bf16 storage)" if amp else "", " [HIP graph]" if use_graph else " [eager]",
      "" if mod.cpp_node else " [Python composition]", e0.elapsed_time(e1) * 1e3 / iters, t_enq * 1e6 / iters))
