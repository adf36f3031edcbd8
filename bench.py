#!/usr/bin/env python3
"""bench.py — MSDeformAttn fwd+bwd samples/s at the Swin-L 4-scale / 300-query shape.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2_decoder] [--no-graph]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one MSDeformAttnFunction.apply forward plus its backward (all three
gradients) over one batch of synthetic inputs already resident in HBM
(BASELINE.json configs[1]: N=2, levels 48/24/12/6 from a 384x384 input, 300 queries,
8 heads, D=32, P=4, fp32).  One sample = one batch element.  With N GPUs every rank
runs the same per-rank batch on its own data (the op shards over the batch with no
exchange), so the scaling is weak and there is no data-path collective;
`value` = samples all ranks processed / max-over-ranks time.

The step is captured once into a HIP graph (two kernel launches + no host work per
replay) and replayed K times; --no-graph times the eager autograd path instead.  The K-step
block (barrier + synchronize on both sides, max over ranks) is repeated --repeats times and the
MEDIAN block is reported, so a short driver run (--steps 20 = 0.4 ms) is not a single sample;
the same line also carries `eager_ms_per_step` (plain autograd calls, what a drop-in user of the
module runs) and `graph1_ms_per_step` (one graph launch per step) next to the headline.
Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: `roofline` (dominant kernel = backward; algorithmic bytes / HIP-event time
against the 8 TB/s HBM peak; `traffic` = HBM bytes per launch from the PMC passes, only
while the kernel sources are the profiled ones) and `cpu_baseline` (the PyTorch-CPU
fallback port timed on the host cores of this node; test infrastructure under oracle/,
never the thing measured as `value`).  At N = 1 the same line carries the rest of
DESIGN.md's performance table, measured in the same run (none of it is `value`):
`workloads` — cfg2_encoder / cfg4_decoder / cfg4_encoder in fp32, cfg2_decoder with
bf16 rows, model-like locations (SURVEY 8d distribution B) at cfg2_decoder / cfg4_encoder and one
row per shape with the deterministic flag: graph-replay step, the two kernels on their own,
algorithmic bytes and fractions — and `modules` — the MSDeformAttn module forward+backward at cfg2_decoder and
cfg4_encoder, fp32 and autocast-bf16, per HIP graph and eager wall time — and `layers` — the decoder self-attention core
(msda_attn32_*) at 300 queries x 32 frames x 8 heads with torch's fused attention beside it (--no-table skips all three).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
METRIC = "MSDeformAttn fwd+bwd samples/sec @ Swin-L 4-scale, 300 queries, 1/2/4/8 MI355X"      # BASELINE.json "metric"

WORKLOADS = {
    # name: (N per GPU, [(H, W)...], M, D, Lq, P)
    "cfg2_decoder": (2, [(48, 48), (24, 24), (12, 12), (6, 6)], 8, 32, 300, 4),
    "cfg2_encoder": (2, [(48, 48), (24, 24), (12, 12), (6, 6)], 8, 32, 3060, 4),
    "cfg4_decoder": (32, [(28, 28), (14, 14), (7, 7), (4, 4)], 8, 32, 300, 4),
    "cfg4_encoder": (32, [(28, 28), (14, 14), (7, 7), (4, 4)], 8, 32, 1045, 4),
}


def algorithmic_bytes(N, S, M, D, L, Lq, P, e=4, el=4):
    """every input read once, every output written once.  e = bytes per element of the
    value-like tensors (value, out, grad_out, grad_value), el = of the location-like ones."""
    fwd = e * (N * S * M * D + N * Lq * M * D) + el * 3 * N * Lq * M * L * P + 8 * 3 * L
    bwd = e * (N * Lq * M * D + 2 * N * S * M * D) + el * 6 * N * Lq * M * L * P + 8 * 3 * L
    return fwd, bwd


def model_like_locations(N, shapes, M, Lq, P, g):
    """SURVEY.md §8d distribution B: what the UVHand transformers actually feed the op.  Decoder regime
    (Lq != S): reference points ~ U(-1, 1) per query (two-stage sigmoid()*2-1, arctic_transformer.py:230),
    so many taps fall outside the maps; encoder regime (Lq == S): the pixel-centre grid (:314-322).  Offsets
    follow the module's initial pattern (modules/ms_deform_attn.py:64-70): point k of head m sits k+1 pixels
    along head m's direction, plus N(0, 0.5 px) jitter."""
    import math
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    if Lq == S:
        refs = []
        for h, w in shapes:
            ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
            refs.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1))
        ref = torch.cat(refs, 0)[None].expand(N, -1, -1)
    else:
        ref = torch.rand(N, Lq, 2, generator=g) * 2 - 1
    theta = torch.arange(M, dtype=torch.float32) * (2 * math.pi / M)
    dirs = torch.stack((theta.cos(), theta.sin()), -1)
    dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]                                  # [M, 2]
    k = torch.arange(1, P + 1, dtype=torch.float32)
    off_px = dirs[:, None, None, :] * k[None, None, :, None]                           # [M, 1, P, 2]
    off_px = off_px.expand(M, L, P, 2) + 0.0
    wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)                # [L, 2]
    jitter = torch.randn(N, Lq, M, L, P, 2, generator=g) * 0.5
    return ref[:, :, None, None, None, :] + (off_px[None, None] + jitter) / wh[None, None, None, :, None, :]


def make_inputs(workload, seed, device, dist_kind="uniform"):
    """Synthetic inputs.  "uniform": the distributions of models/ops/test.py:33-36 (SURVEY.md §8d,
    distribution A, the headline); "model": distribution B (model_like_locations)."""
    N, shapes, M, D, Lq, P = WORKLOADS[workload]
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes_t = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shapes_t.new_zeros(1), shapes_t.prod(1).cumsum(0)[:-1]))
    S, L = int(shapes_t.prod(1).sum()), len(shapes)
    value = torch.rand(N, S, M, D, generator=g) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    if dist_kind == "model":
        loc = model_like_locations(N, shapes, M, Lq, P, g).contiguous()
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = attn / attn.sum((-1, -2), keepdim=True)
    go = torch.rand(N, Lq, M * D, generator=g)
    host = dict(value=value, shapes=shapes_t, lsi=lsi, loc=loc, attn=attn, go=go)
    dev = {k: v.to(device) for k, v in host.items()} if device is not None else None
    return host, dev, (N, S, M, D, L, Lq, P)


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def copy_bandwidth_gbs(device, nbytes=1 << 30, iters=10):
    """Achieved device-to-device copy rate (read + write bytes / time): the practical HBM ceiling
    on this box, next to the 8 TB/s spec the roofline fraction is quoted against."""
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dst.copy_(src)
    e1.record()
    e1.synchronize()
    return 2.0 * nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9


def time_cpu_baseline(workload, budget_s=8.0):
    """The reference's CPU comparator (grid_sample fallback; port under oracle/) on this node's host
    cores.  Two thread settings: every usable core, capped at the 16-core share
    a 1-GPU box grants, and 8 (the survey container's setting); the faster one is reported."""
    from oracle.torch_fallback import fwd_bwd
    host, _, dims = make_inputs(workload, 0, None)
    N = dims[0]
    runs = []
    for threads in sorted({min(16, usable_cores()), min(8, usable_cores())}, reverse=True):
        torch.set_num_threads(threads)
        for _ in range(2):
            fwd_bwd(host["value"], host["shapes"], host["loc"], host["attn"], host["go"])
        n, t0 = 0, time.perf_counter()
        while True:
            fwd_bwd(host["value"], host["shapes"], host["loc"], host["attn"], host["go"])
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 400:
                break
        runs.append({"threads": threads, "calls": n, "ms_per_step": 1e3 * el / n, "samples_per_s": N * n / el})
    best = max(runs, key=lambda r: r["samples_per_s"])
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "port",
            "ms_per_step": best["ms_per_step"], "runs": runs, "host_cpus": os.cpu_count(),
            "cpu_model": cpu_model_name(),
            "sample": "%d fwd+bwd calls of the %s batch (N=%d) through oracle/torch_fallback.py "
                      "(restatement of ms_deform_attn_core_pytorch, fp32, %d intra-op threads)"
                      % (best["calls"], workload, N, best["threads"])}


def kernel_sources_sha16():
    """Fingerprint of the sources of the kernels the bench workloads launch (the D = 32 family: msda_d32*.hip / .h and the
    shared device header under uvhand_amd/csrc, names and contents): PMC traffic figures are only valid for the kernels they
    were profiled on.  The generic family (other D, fp64), the projections' GEMM / LayerNorm / flatten kernels are not part
    of these launches and do not enter."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "uvhand_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h")) and (name.startswith("msda_d32") or name == "msda_common.h"):
            h.update(name.encode() + b"\0")
            with open(os.path.join(src, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, which):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs by tools/profile_gpu.sh, corrected as MI355X_MICROARCH.md prescribes: KiB units, FETCH_SIZE doubled
    on gfx950; tools/pmc_traffic.py turns the summaries into profiles/pmc_traffic.json and stamps it with the fingerprint
    of the kernel sources it profiled).  Returns (bytes or None, the fingerprint the figure belongs to): None when the
    workload has not been profiled OR the kernels have changed since — a stale constant is not a measurement."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        at = table.get("sources_sha16")
        val = table[workload][which]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None, None
    return (val if at == kernel_sources_sha16() else None), at


def event_time_ms(fn, iters, stream):
    """Average duration of fn() over `iters` back-to-back calls, HIP events on `stream`."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(stream)
    for _ in range(iters):
        fn()
    stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) / iters


def graph_of(fn, stream, per=10):
    """fn() captured `per` times into one HIP graph (None if capture fails); the caller is inside `with torch.cuda.stream(stream)`."""
    fn(); stream.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(per):
                fn()
        return g
    except Exception as exc:
        print("bench: HIP graph capture failed (%s)" % exc, file=sys.stderr)
        return None


def timed_us(call, per, stream, budget_s=0.6):
    """Device time per inner call: HIP events on `stream` around enough back-to-back `call()`s (each = `per` inner calls)
    to fill ~budget_s, after a short warm-up."""
    for _ in range(3):
        call()
    one = event_time_ms(call, 2, stream)
    n = max(3, min(2000, int(budget_s * 1e3 / max(one, 1e-3))))
    return 1e3 * event_time_ms(call, n, stream) / per


def measure_op(workload, dtype, device, stream, seed, locations="uniform", budget_s=0.6, deterministic=False):
    """One extra row of the performance table (not `value`): the same step as the headline — Function.apply forward +
    backward, 10 steps per HIP graph — on another workload / storage type / location distribution ("model" =
    distribution B) / with the deterministic flag, plus the two kernels on their own."""
    from uvhand_amd import _native
    from uvhand_amd.functions import MSDeformAttnBF16Function, MSDeformAttnFunction
    bf16 = dtype == "bf16"
    _, d, dims = make_inputs(workload, seed, device, locations)
    N, S, M, D, L, Lq, P = dims
    value = (d["value"].to(torch.bfloat16) if bf16 else d["value"]).requires_grad_(True)
    loc, attn = d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
    go = d["go"].to(torch.bfloat16) if bf16 else d["go"]
    apply = MSDeformAttnBF16Function.apply if bf16 else MSDeformAttnFunction.apply

    def step():
        value.grad = loc.grad = attn.grad = None
        apply(value, d["shapes"], d["lsi"], loc, attn, 64).backward(go)

    # (the Functions read the switch when they run: MSDA_DETERMINISTIC=1 for the life of this row's captures)
    prev_det = os.environ.get("MSDA_DETERMINISTIC")
    if deterministic:
        os.environ["MSDA_DETERMINISTIC"] = "1"
    vd, ld, ad = value.detach(), loc.detach(), attn.detach()
    gv32 = bf16 and _native.backward_passes(Lq, P) > 1
    # the two kernels as the autograd step runs them: the forward leaves its point table where the backward's plan reads one
    table = _native.ms_deform_attn_forward(vd, d["shapes"], d["lsi"], ld, ad, 64, with_table=True)[1]
    fwd = lambda: _native.ms_deform_attn_forward(vd, d["shapes"], d["lsi"], ld, ad, 64, with_table=True if table is not None else None)
    bwd = lambda: _native.ms_deform_attn_backward(vd, d["shapes"], d["lsi"], ld, ad, go, 64, fp32_grad_value=gv32, table=table,
                                                  deterministic=deterministic)
    row = {"workload": workload, "dtype": dtype, "locations": locations, "deterministic": bool(deterministic), "N": N, "Lq": Lq, "S": S}
    try:
        with torch.cuda.stream(stream):
            for name, fn in (("step", step), ("fwd", fwd), ("bwd", bwd)):
                g = graph_of(fn, stream)
                call, per = (g.replay, 10) if g is not None else (fn, 1)
                # the better of two timed batches: single batches of these side rows were seen 20 % off (a 14.4 / 17.4 us pair
                # for one step on two runs of the same code) while the kernels' own times agreed to 1 %
                row[name + "_us"] = min(timed_us(call, per, stream, budget_s / 2), timed_us(call, per, stream, budget_s / 2))
                del g
            stream.synchronize()
    finally:
        if deterministic:
            if prev_det is None:
                os.environ.pop("MSDA_DETERMINISTIC", None)
            else:
                os.environ["MSDA_DETERMINISTIC"] = prev_det
    fwd_b, bwd_b = algorithmic_bytes(N, S, M, D, L, Lq, P, e=2 if bf16 else 4)
    row.update({"ms_per_step": row["step_us"] * 1e-3, "samples_per_s": N / (row["step_us"] * 1e-6),
                "fwd_algorithmic_bytes": fwd_b, "bwd_algorithmic_bytes": bwd_b,
                "fwd_frac": fwd_b / (row["fwd_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "bwd_frac": bwd_b / (row["bwd_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "launch": "hipGraph replay, 10 steps per graph; HIP events"})
    tr, at = pmc_traffic(workload + ("_bf16" if bf16 else ""), "bwd") if (locations == "uniform" and not deterministic) else (None, None)
    row["bwd_traffic"] = tr
    row["bwd_traffic_ratio"] = tr / bwd_b if tr else None
    row["forward_table_bytes"] = int(table.numel()) if table is not None else 0
    return row


def measure_module(workload, device, stream, amp=False, budget_s=0.6):
    """MSDeformAttn MODULE forward + backward (the op, its four nn.Linear projections, softmax / location arithmetic and
    the weight gradients): one step per HIP graph (GPU-bound time) and the eager wall time a drop-in user sees."""
    from uvhand_amd.modules import MSDeformAttn
    N, shapes, M, D, Lq, P = WORKLOADS[workload]
    gen = torch.Generator(device="cpu").manual_seed(0)
    torch.manual_seed(0)
    mod = MSDeformAttn(M * D, len(shapes), M, P).to(device)
    with torch.no_grad():
        for p in mod.parameters():
            p.add_(torch.randn(p.shape, generator=gen).to(device) * 0.02)
    mod.bf16_storage = bool(amp)
    sh = torch.tensor(shapes, dtype=torch.long, device=device)
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    S = int(sh.prod(1).sum())
    q = torch.randn(N, Lq, M * D, generator=gen).to(device).requires_grad_(True)
    src = torch.randn(N, S, M * D, generator=gen).to(device).requires_grad_(True)
    ref = torch.rand(N, Lq, len(shapes), 2, generator=gen).to(device).requires_grad_(True)
    go = torch.randn(N, Lq, M * D, generator=gen).to(device)

    def step():
        mod.zero_grad(set_to_none=True)
        q.grad = src.grad = ref.grad = None
        if amp:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = mod(q, ref, src, sh, lsi)
            out.backward(go.to(out.dtype))
        else:
            mod(q, ref, src, sh, lsi).backward(go)

    row = {"workload": workload, "amp": "bf16" if amp else None}
    with torch.cuda.stream(stream):
        for _ in range(3):
            step()
        g = graph_of(step, stream, per=1)
        if g is not None:
            row["graph_us"] = timed_us(g.replay, 1, stream, budget_s)
        # eager: wall clock around back-to-back calls (host-bound at decoder sizes)
        for _ in range(5):
            step()
        stream.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        stream.synchronize()
        row["eager_wall_us"] = 1e6 * (time.perf_counter() - t0) / n
    return row


def measure_attention(device, stream, budget_s=0.4, L=300, N=32, H=8, p=0.1):
    """the decoder self-attention core (msda_attn32_*_f32, head_dim 32) at the training shape — 300 queries, 32 frames,
    8 heads, dropout 0.1 — forward and backward per HIP graph of 10 calls, with torch's scaled_dot_product_attention beside it."""
    import math
    import torch.nn.functional as F
    from uvhand_amd import _native
    E = H * 32
    gen = torch.Generator(device="cpu").manual_seed(1)
    qk = torch.randn(L, N, 2 * E, generator=gen).to(device)
    v, go = torch.randn(L, N, E, generator=gen).to(device), torch.randn(L, N, E, generator=gen).to(device)
    q, k = qk[..., :E], qk[..., E:]
    seed = torch.tensor([20240607], dtype=torch.int64, device=device)
    scale = 1.0 / math.sqrt(32)
    row = {"what": "decoder self-attention core", "queries": L, "frames": N, "heads": H, "head_dim": 32, "dropout": p, "dtype": "f32"}
    with torch.cuda.stream(stream):
        out, lse = _native.attn32_forward(q, k, v, H, scale, p, seed)
        gqk, gv = torch.empty_like(qk), torch.empty_like(v)
        fwd = lambda: _native.attn32_forward(q, k, v, H, scale, p, seed)
        bwd = lambda: _native.attn32_backward(q, k, v, out, lse, go, H, scale, p, seed, grad_q=gqk[..., :E], grad_k=gqk[..., E:], grad_v=gv)
        for name, fn in (("fwd_us", fwd), ("bwd_us", bwd)):
            fn()
            g = graph_of(fn, stream, per=10)
            row[name] = timed_us(g.replay, 10, stream, budget_s) if g is not None else None
        q4, k4, v4 = (t.contiguous().view(L, N * H, 32).transpose(0, 1).reshape(N, H, L, 32).detach().requires_grad_(True) for t in (q, k, v))
        g4 = go.view(L, N * H, 32).transpose(0, 1).reshape(N, H, L, 32).contiguous()

        def torch_step():
            q4.grad = k4.grad = v4.grad = None
            F.scaled_dot_product_attention(q4, k4, v4, dropout_p=p).backward(g4)
        torch_step()
        g = graph_of(torch_step, stream, per=10)
        row["torch_sdpa_fwd_bwd_us"] = timed_us(g.replay, 10, stream, budget_s) if g is not None else None
    flops = 4.0 * N * H * L * L * 32
    if row.get("fwd_us") and row.get("bwd_us"):
        row["fwd_bwd_us"] = row["fwd_us"] + row["bwd_us"]
        row["fwd_tflops"], row["bwd_tflops"] = flops / row["fwd_us"] / 1e6, 2.5 * flops / row["bwd_us"] / 1e6
        row["mfma_f32_peak_tflops"] = 157.3
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg2_decoder", choices=sorted(WORKLOADS))
    ap.add_argument("--no-graph", action="store_true", help="time the eager autograd path")
    ap.add_argument("--graph-steps", type=int, default=10,
                    help="steps captured per HIP graph (the timed loop replays it steps/graph-steps times; "
                         "reduced to a divisor of --steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-table", action="store_true",
                    help="skip the `workloads` / `modules` arrays (the other shapes of DESIGN.md's performance table, ~2 s each)")
    ap.add_argument("--repeats", type=int, default=50,
                    help="repetitions of the timed --steps block (each bracketed by barrier + synchronize); the median is reported")
    ap.add_argument("--kernel-iters", type=int, default=200)
    ap.add_argument("--locations", default="uniform", choices=["uniform", "model"],
                    help="sampling-location distribution: uniform in [0,1) (headline) or model-like (SURVEY §8d B)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="storage type of value/out/grad tensors (bf16 = BASELINE config 3, MSDeformAttnBF16Function)")
    args = ap.parse_args()

    from uvhand_amd import harness
    rank, local_rank, world = harness.dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    import torch.distributed as dist
    # one rank per GPU; MSDA_BENCH_BACKEND=gloo lets several ranks rehearse the N>1 path on one card
    backend = os.environ.get("MSDA_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > 1 and local_rank >= n_dev:
        sys.exit("bench.py: LOCAL_RANK %d but only %d visible GPU(s): RCCL needs one rank per device "
                 "(MSDA_BENCH_BACKEND=gloo rehearses several ranks on one card)" % (local_rank, n_dev))
    dev_index = local_rank % max(1, n_dev)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    harness.init_process_group(backend, device)
    identities = harness.check_one_rank_per_device(backend if world > 1 else "none", device)   # raises under nccl
    runtime = harness.collective_runtime()
    if rank == 0:
        print("bench: %s" % json.dumps(runtime), file=sys.stderr, flush=True)

    from uvhand_amd import _native
    from uvhand_amd.functions import MSDeformAttnBF16Function, MSDeformAttnFunction
    _native.load()
    bf16 = args.dtype == "bf16"
    esize = 2 if bf16 else 4
    fn_apply = MSDeformAttnBF16Function.apply if bf16 else MSDeformAttnFunction.apply

    _, d, dims = make_inputs(args.workload, harness.rank_seed(1000, rank), device, args.locations)
    N, S, M, D, L, Lq, P = dims
    value = (d["value"].to(torch.bfloat16) if bf16 else d["value"]).requires_grad_(True)
    loc = d["loc"].requires_grad_(True)
    attn = d["attn"].requires_grad_(True)
    shapes, lsi = d["shapes"], d["lsi"]
    go = d["go"].to(torch.bfloat16) if bf16 else d["go"]

    def step():
        value.grad = loc.grad = attn.grad = None
        out = fn_apply(value, shapes, lsi, loc, attn, 64)
        out.backward(go)

    stream = torch.cuda.Stream(device)
    torch.cuda.synchronize()
    graph = None
    with torch.cuda.stream(stream):
        for _ in range(3):
            step()
        stream.synchronize()
        per_graph = 1
        if not args.no_graph:
            per_graph = max(1, min(args.graph_steps, args.steps))
            while args.steps % per_graph:                 # exactly --steps steps are timed
                per_graph -= 1
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    for _ in range(per_graph):            # stream order chains them: no two steps overlap
                        step()
            except Exception as exc:                      # report, then fall back to eager timing
                print("bench: HIP graph capture failed (%s); timing eager" % exc, file=sys.stderr)
                graph, per_graph = None, 1
        run = graph.replay if graph is not None else step

        def barrier():
            harness.barrier(device)

        def timed_block(call, calls):
            """EXACTLY `calls` invocations bracketed by barrier + synchronize on both sides.  Returns this rank's seconds
            from the common start (every rank past the barrier, device idle) to ITS OWN completion (device idle again);
            the closing barrier follows the clock read, so the collective's own latency (tens of us on RCCL, a
            sizeable share of a 20-step block) is not billed to the steps — the slowest rank's time, taken by the
            caller as the max over ranks, is when the whole job was done."""
            barrier()
            t0 = time.perf_counter()
            for _ in range(calls):
                call()
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            barrier()
            return t1 - t0

        for _ in range((args.warmup + per_graph - 1) // per_graph):
            run()
        # --repeats blocks, but no more than fit in ~5 s of timed work (the first block tells how long one takes)
        blocks = [timed_block(run, args.steps // per_graph)]
        fit = int(5.0 / max(blocks[0], 1e-6))
        n_blocks = max(1, min(args.repeats, max(3, fit)))
        n_blocks = int(harness.max_over_ranks(n_blocks, device) + 0.5) if world > 1 else n_blocks    # same count on every rank
        blocks += [timed_block(run, args.steps // per_graph) for _ in range(n_blocks - 1)]

        # the other two launch modes of the same step, for the record (not `value`)
        side = {}
        if graph is not None and world == 1:
            for _ in range(min(args.warmup, 20)):
                step()
            n_eager = min(args.steps, 200)
            side["eager_ms_per_step"] = 1e3 * sorted(timed_block(step, n_eager) for _ in range(5))[2] / n_eager
            try:
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1, stream=stream):
                    step()
                for _ in range(10):
                    g1.replay()
                n_g1 = min(args.steps, 200)
                side["graph1_ms_per_step"] = 1e3 * sorted(timed_block(g1.replay, n_g1) for _ in range(5))[2] / n_g1
            except Exception as exc:
                print("bench: single-step graph failed (%s)" % exc, file=sys.stderr)

        # ---- per-kernel timing for the roofline (HIP events on the launch stream) ----
        vd, ld, ad = value.detach(), loc.detach(), attn.detach()
        if bf16:
            ld, ad = ld.float(), ad.float()
        # (as the autograd step runs them: the forward leaves its point table where the backward's plan reads one)
        table = _native.ms_deform_attn_forward(vd, shapes, lsi, ld, ad, 64, with_table=True)[1]
        fwd = lambda: _native.ms_deform_attn_forward(vd, shapes, lsi, ld, ad, 64, with_table=True if table is not None else None)
        # bf16 rows: the kernel variant MSDeformAttnBF16Function picks for a bf16 `value` (fp32 grad_value
        # when the backward takes several passes)
        gv32 = bf16 and _native.backward_passes(Lq, P) > 1
        bwd = lambda: _native.ms_deform_attn_backward(vd, shapes, lsi, ld, ad, go, 64, fp32_grad_value=gv32, table=table)
        kt = {}
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            g2 = None
            try:
                fn(); stream.synchronize()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=stream):
                    for _ in range(10):
                        fn()
                call, per = g2.replay, 10
            except Exception:
                call, per = fn, 1
            for _ in range(5):
                call()
            kt[name] = event_time_ms(call, max(1, args.kernel_iters // per), stream) / per
        stream.synchronize()

    total_samples = harness.sum_over_ranks(N * args.steps, device)
    # every block: the slowest rank's time (what the whole job waited for); then the median block
    block_max = sorted(harness.max_over_ranks(t, device) for t in blocks)
    elapsed = block_max[len(block_max) // 2]
    ranks_seen = harness.gather_objects({"rank": rank, "local_rank": local_rank, "device": "cuda:%d" % dev_index,
                                         "device_name": torch.cuda.get_device_name(dev_index), "pid": os.getpid(),
                                         "pci_bus_id": identities[rank]["pci_bus_id"], "uuid": identities[rank]["uuid"],
                                         "seed": harness.rank_seed(1000, rank), "samples": N * args.steps,
                                         "median_block_s": sorted(blocks)[len(blocks) // 2]})
    print("bench: rank %d/%d backend=%s device=cuda:%d (%s, pci %s) pid=%d" % (
        rank, world, backend if world > 1 else "none", dev_index, torch.cuda.get_device_name(dev_index),
        identities[rank]["pci_bus_id"], os.getpid()), file=sys.stderr, flush=True)

    if rank == 0:
        fwd_b, bwd_b = algorithmic_bytes(N, S, M, D, L, Lq, P, e=esize)
        ach = bwd_b / (kt["bwd"] * 1e-3) / 1e9
        traffic, traffic_at = pmc_traffic(args.workload + ("_bf16" if bf16 else ""), "bwd")
        result = {
            "metric": METRIC,
            "value": total_samples / elapsed,
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: N=%d/GPU, levels %s (S=%d), Lq=%d, M=%d, D=%d, P=%d, %s"
                                   % (args.workload, N, "/".join(str(h) for h, _ in WORKLOADS[args.workload][1]),
                                      S, Lq, M, D, P, "fp32" if not bf16 else "bf16 storage / fp32 accumulate"),
                       "locations": args.locations,
                       "step": "MSDeformAttnFunction.apply forward + backward (3 grads)",
                       "launch": ("hipGraph replay, %d step(s) per graph" % per_graph) if graph is not None
                                 else "eager autograd",
                       # the same step launched the way the reference's training loop does (plain autograd calls, no graph)
                       # and as one graph launch per step: NOT `value`, here so that the launch mode cannot be missed
                       "eager_ms_per_step": side.get("eager_ms_per_step"),
                       "graph1_ms_per_step": side.get("graph1_ms_per_step"),
                       "sharding": "batch-sharded, no collective",
                       "timing": "median of %d blocks of %d steps, each bracketed by barrier + synchronize; max over ranks "
                                 "per block (min %.4f / max %.4f ms per step)" % (len(block_max), args.steps,
                                                                                  1e3 * block_max[0] / args.steps,
                                                                                  1e3 * block_max[-1] / args.steps),
                       "backend": backend if world > 1 else None},
            "ranks": ranks_seen, "runtime": runtime,
            "roofline": {"bound": "hbm", "kernel": "backward: msda::bwd_fused_d32_kernel (grad_value sort+gather "
                                                    "workgroups and grad_loc/grad_attn workgroups in one launch)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_ratio": (traffic / bwd_b) if traffic else None,
                         "traffic_profiled_at": traffic_at, "sources_sha16": kernel_sources_sha16(),
                         "algorithmic_bytes": bwd_b, "ms": kt["bwd"]},
            "kernels": {"fwd": {"ms": kt["fwd"], "algorithmic_bytes": fwd_b,
                                "GBps": fwd_b / (kt["fwd"] * 1e-3) / 1e9},
                        "bwd": {"ms": kt["bwd"], "algorithmic_bytes": bwd_b, "GBps": ach}},
        }
        result.update(side)
        # The HBM fraction above is a CACHE-RESIDENT figure: the timed loop replays the same buffers, and the whole working set
        # (fwd + bwd tensors, ~23 MB at cfg-2) sits in the 8 L2s / the 256 MB Infinity Cache (BASELINE.md says to state this).
        working_set = fwd_b + bwd_b - esize * (N * S * M * D + N * Lq * M * D) - 4 * 3 * N * Lq * M * L * P   # inputs counted once
        result["roofline"]["cache_resident"] = True
        result["roofline"]["forward_table_bytes"] = int(table.numel()) if table is not None else 0   # extra (not algorithmic) traffic: written by the forward, read by the backward
        result["roofline"]["working_set_bytes"] = working_set
        if world == 1:
            result["roofline"]["copy_GBps_measured"] = copy_bandwidth_gbs(device)
            # second ceiling: on cache-resident data a gather kernel is bound by the ROW REQUESTS the vector memory path serves,
            # not by HBM bytes.  Measured live (msda_probe_row_gather: the kernels' access pattern and nothing else) on a table
            # of the size of `value`; achieved = the tap rows the backward has to gather (4 per sampling point for grad_loc /
            # grad_attn plus 4 per point for grad_value) / its time.
            try:
                with torch.cuda.stream(stream):
                    ceiling = _native.probe_row_gather(esize * N * S * M * D, device, row_bytes=128 if esize == 4 else 64)
                    stream.synchronize()
                tap_rows = 2 * 4 * N * Lq * M * L * P
                ach_rows = tap_rows / (kt["bwd"] * 1e-3)
                result["roofline"]["secondary"] = {"bound": "tap rows/s (vector memory path, cache-resident table)",
                                                   "achieved": ach_rows, "ceiling": ceiling, "frac": ach_rows / ceiling,
                                                   "unit": "rows/s", "rows_per_launch": tap_rows,
                                                   "ceiling_source": "msda_probe_row_gather on a %d-byte table, same box, same run" % (esize * N * S * M * D)}
            except Exception as exc:
                result["roofline"]["secondary"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if world == 1 and not args.no_table:
            # the rest of DESIGN.md's performance table, measured in the same run (none of it is `value`)
            seed = harness.rank_seed(1000, rank)
            table = [(w, "f32", "uniform", False) for w in WORKLOADS if (w, "f32") != (args.workload, args.dtype)]
            table += [("cfg2_decoder", "bf16", "uniform", False)] if (args.workload, args.dtype) != ("cfg2_decoder", "bf16") else []
            # SURVEY 8d distribution B (model-like locations) on the headline shape and the training encoder shape, and the
            # deterministic flag's cost on every shape
            table += [("cfg2_decoder", "f32", "model", False), ("cfg4_encoder", "f32", "model", False)]
            table += [(w, "f32", "uniform", True) for w in WORKLOADS]
            result["workloads"] = []
            for w, dt, locs, det in table:
                try:
                    result["workloads"].append(measure_op(w, dt, device, stream, seed, locs, 0.4 if (det or locs != "uniform") else 0.6, det))
                except Exception as exc:                  # a row that cannot be measured says so; the headline stands
                    result["workloads"].append({"workload": w, "dtype": dt, "locations": locs, "deterministic": det,
                                                "error": "%s: %s" % (type(exc).__name__, exc)})
                torch.cuda.empty_cache()
            result["modules"] = []
            for w, amp in (("cfg2_decoder", False), ("cfg4_encoder", False), ("cfg2_decoder", True), ("cfg4_encoder", True)):
                try:
                    result["modules"].append(measure_module(w, device, stream, amp))
                except Exception as exc:
                    result["modules"].append({"workload": w, "amp": amp, "error": "%s: %s" % (type(exc).__name__, exc)})
                torch.cuda.empty_cache()
            try:
                result["layers"] = [measure_attention(device, stream)]
            except Exception as exc:
                result["layers"] = [{"what": "decoder self-attention core", "error": "%s: %s" % (type(exc).__name__, exc)}]
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = time_cpu_baseline(args.workload)
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
