#!/usr/bin/env python3
"""Where the 12-layer train step's wall time goes on the host: enqueue time of forward / backward / optimizer and the time the
host then waits for the GPU, with the attention modules as C++ nodes (default) or as the Python composition (--no-cpp-node)."""
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
for name, m in model.named_modules():
    if isinstance(m, MSDeformAttn):
        m.cpp_node = cpp
        if "--cpp-enc-only" in sys.argv:
            m.cpp_node = name.startswith("enc.")
        if "--cpp-dec-only" in sys.argv:
            m.cpp_node = name.startswith("dec.")
opt = torch.optim.AdamW(model.parameters(), lr=2e-5, weight_decay=1e-4)
shapes_list = [(28, 28), (14, 14), (7, 7), (4, 4)]
shapes = torch.tensor(shapes_list, dtype=torch.long, device=dev)
lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
S = int(shapes.prod(1).sum()); W = 32
src = torch.randn(W, S, 256, device=dev); pos = torch.randn(W, S, 256, device=dev) * 0.1
ref = encoder_reference_points(shapes_list, torch.ones(W, 4, 2, device=dev), dev)
import gc
if "--no-gc" in sys.argv:
    gc.disable()
if "--gc-freeze" in sys.argv:
    gc.collect(); gc.freeze()
acc = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0, "wait": 0.0, "g_fwd": 0.0, "g_bwd": 0.0, "g_opt": 0.0}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
stats0 = None
for it in range(25):
    if it == 5:
        stats0 = torch.cuda.memory_stats(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev[0].record()
    opt.zero_grad(set_to_none=True)
    loss = model(src, pos, ref, shapes, lsi)
    ev[1].record()
    t1 = time.perf_counter()
    loss.backward()
    ev[2].record()
    t2 = time.perf_counter()
    if "--sync-after-backward" in sys.argv:
        torch.cuda.synchronize()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1); opt.step()
    ev[3].record()
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if it >= 5:
        acc["fwd"] += t1 - t0; acc["bwd"] += t2 - t1; acc["opt"] += t3 - t2; acc["wait"] += t4 - t3
        acc["g_fwd"] += ev[0].elapsed_time(ev[1]) * 1e-3; acc["g_bwd"] += ev[1].elapsed_time(ev[2]) * 1e-3
        acc["g_opt"] += ev[2].elapsed_time(ev[3]) * 1e-3
stats1 = torch.cuda.memory_stats(dev)
print("  allocator over the 20 timed steps: device allocs %d, frees %d, retries %d; peak allocated %.0f MB, reserved %.0f MB" % (
    stats1["num_device_alloc"] - stats0["num_device_alloc"], stats1["num_device_free"] - stats0["num_device_free"],
    stats1["num_alloc_retries"] - stats0["num_alloc_retries"], stats1["allocated_bytes.all.peak"] / 1e6, stats1["reserved_bytes.all.peak"] / 1e6))
print("%s: forward enqueue %.2f ms, backward %.2f ms, clip+optimizer %.2f ms, then waiting for the GPU %.2f ms; total %.2f ms/step" % (
    "C++ module nodes" if cpp else "Python composition", *(1e3 * acc[k] / 20 for k in ("fwd", "bwd", "opt", "wait")),
    1e3 * sum(acc[k] for k in ("fwd", "bwd", "opt", "wait")) / 20))
print("  on the GPU's own clock (events at the phase boundaries): forward %.2f ms, backward %.2f ms, clip+optimizer %.2f ms" % (
    1e3 * acc["g_fwd"] / 20, 1e3 * acc["g_bwd"] / 20, 1e3 * acc["g_opt"] / 20))
if "--profile-steps" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            model(src, pos, ref, shapes, lsi).backward()
            torch.cuda.synchronize()
            with torch.profiler.record_function("CLIP"):
                torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
            with torch.profiler.record_function("OPT"):
                opt.step()
            torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=45, max_name_column_width=50))
if "--profile" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    opt.zero_grad(set_to_none=True)
    model(src, pos, ref, shapes, lsi).backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1); opt.step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
    kinds = {}
    for p in model.parameters():
        g = p.grad
        k = (g.is_contiguous(), g.storage_offset() != 0, g.untyped_storage().nbytes() != g.numel() * 4, str(g.dtype))
        kinds[k] = kinds.get(k, 0) + 1
    print("grad kinds (contiguous, offset != 0, storage larger than the tensor, dtype):", kinds)
