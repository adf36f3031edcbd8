#!/usr/bin/env python3
"""Diagnostic: which threads burn CPU while the main thread runs clip + optimizer after a backward through the C++ module
nodes (default) or the Python composition (--no-cpp-node).  GPU drained before the phase, so only host work is timed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from ddp_step import SyntheticDeformableStack
from uvhand_amd.modules import MSDeformAttn
from uvhand_amd.utils import encoder_reference_points
cpp = "--no-cpp-node" not in sys.argv
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = SyntheticDeformableStack(6, 6, 300, 0.0, dropout=0.1).to(dev)
for m in model.modules():
    if isinstance(m, MSDeformAttn):
        m.cpp_node = cpp
opt = torch.optim.AdamW(model.parameters(), lr=2e-5, weight_decay=1e-4)
shapes_list = [(28, 28), (14, 14), (7, 7), (4, 4)]
shapes = torch.tensor(shapes_list, dtype=torch.long, device=dev)
lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
S = int(shapes.prod(1).sum()); W = 32
src = torch.randn(W, S, 256, device=dev); pos = torch.randn(W, S, 256, device=dev) * 0.1
ref = encoder_reference_points(shapes_list, torch.ones(W, 4, 2, device=dev), dev)


def task_times():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read().rsplit(")", 1)[1].split()
            out[int(tid)] = (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK")
        except OSError:
            pass
    return out


import gc
gc_log = {"n": 0, "t": 0.0, "t0": 0.0, "in_phase": 0.0, "phase": False}
def _gc_cb(phase, info):
    if phase == "start":
        gc_log["t0"] = time.perf_counter()
    else:
        dt = time.perf_counter() - gc_log["t0"]; gc_log["n"] += 1; gc_log["t"] += dt
        if gc_log["phase"]:
            gc_log["in_phase"] += dt
gc.callbacks.append(_gc_cb)
if "--no-gc" in sys.argv:
    gc.disable()
wall = {"clip": 0.0, "opt": 0.0}; cpu = {"clip": 0.0, "opt": 0.0}; others = 0.0
for it in range(25):
    opt.zero_grad(set_to_none=True)
    model(src, pos, ref, shapes, lsi).backward()
    torch.cuda.synchronize()
    if "--sleep" in sys.argv:
        time.sleep(0.3)
    before = task_times(); gc_log["phase"] = True
    w0, c0 = time.perf_counter(), time.thread_time()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
    w1, c1 = time.perf_counter(), time.thread_time()
    opt.step()
    w2, c2 = time.perf_counter(), time.thread_time()
    after = task_times(); gc_log["phase"] = False
    torch.cuda.synchronize()
    if it >= 5:
        wall["clip"] += w1 - w0; wall["opt"] += w2 - w1; cpu["clip"] += c1 - c0; cpu["opt"] += c2 - c1
        others += sum(after[t] - before.get(t, 0.0) for t in after) - (c2 - c0)
# the pieces of clip_grad_norm_, one by one, on the last step's gradients
def piece(label, fn, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        r = fn()
    dt = (time.perf_counter() - t) / n; torch.cuda.synchronize()
    print("   %-40s %.3f ms" % (label, 1e3 * dt)); return r
params = list(model.parameters())
grads = piece("[p.grad for p in params]", lambda: [p.grad for p in params if p.grad is not None])
norms = piece("_foreach_norm", lambda: torch._foreach_norm(grads, 2.0))
tot = piece("stack + vector_norm", lambda: torch.linalg.vector_norm(torch.stack(norms), 2.0))
coef = torch.clamp(0.1 / (tot + 1e-6), max=1.0)
piece("_foreach_mul_(grads, tensor)", lambda: torch._foreach_mul_(grads, coef))
piece("clip_grad_norm_", lambda: torch.nn.utils.clip_grad_norm_(params, 0.1))
piece("clip_grad_norm_(foreach=True)", lambda: torch.nn.utils.clip_grad_norm_(params, 0.1, foreach=True))
piece("clip_grad_norm_ on a generator", lambda: torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1))
kinds = {}
for g in grads:
    k = (type(g).__name__, g.is_contiguous(), g.storage_offset() != 0, g.untyped_storage().nbytes() != g.numel() * 4, str(g.dtype), g.requires_grad)
    kinds[k] = kinds.get(k, 0) + 1
print("   grad kinds:", kinds)
print("   gc: %d collections, %.2f ms in total, %.2f ms inside clip+optimizer (25 steps)" % (gc_log["n"], 1e3 * gc_log["t"], 1e3 * gc_log["in_phase"]))
n_grads = sum(p.grad is not None for p in model.parameters())
print("%s: clip wall %.2f ms (cpu %.2f), optimizer wall %.2f ms (cpu %.2f); other threads' cpu during the phase %.2f ms; "
      "threads %d, torch threads %d, grads %d, cores %d" % (
          "C++ nodes" if cpp else "Python", 1e3 * wall["clip"] / 20, 1e3 * cpu["clip"] / 20, 1e3 * wall["opt"] / 20, 1e3 * cpu["opt"] / 20,
          1e3 * others / 20, len(os.listdir("/proc/self/task")), torch.get_num_threads(), n_grads, len(os.sched_getaffinity(0))))
