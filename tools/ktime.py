#!/usr/bin/env python3
"""Per-call device time of the native forward / backward (HIP events, graph of 10 calls)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_inputs, algorithmic_bytes, WORKLOADS
from uvhand_amd import _native
if any(k.startswith("MSDA_") for k in os.environ):          # A/B knobs live in the diagnostic build only
    _native.LIB_PATH = os.path.join(ROOT, "uvhand_amd", "libmsda_hip_tuning.so")
if os.environ.get("KTIME_LIB"):                              # a library built with other compile-time constants (make OUT=... EXTRA=-D...)
    _native.LIB_PATH = os.path.join(ROOT, os.environ["KTIME_LIB"])

PYRAMIDS = {"p48": [(48, 48), (24, 24), (12, 12), (6, 6)],          # BASELINE configs[1]: Swin-L 4-scale from 384 x 384
            "p28": [(28, 28), (14, 14), (7, 7), (4, 4)],            # configs[3]: the per-rank training shape
            "p40x3": [(40, 40), (20, 20), (10, 10)]}                # a 3-level pyramid (no bench shape has one)


def time_call(fn, st):
    fn(); st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(10):
            fn()
    for _ in range(3):
        g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(20):
        g.replay()
    b.record(st); b.synchronize()
    return a.elapsed_time(b) * 1e3 / 200


def sweep(out_path):
    """--sweep <out.json>: N in {1..64} x {Lq = 300, Lq = S} x three pyramids x {uniform, model-like} locations — forward and
    backward time per call (forward table in use, as under autograd), samples/s, fraction of the HBM roofline by algorithmic
    bytes, and the launch plan the library chose (msda_describe_plan).  Looks for dispatcher cliffs: the report at the end
    lists every place where doubling N loses more than 15 % of the samples/s."""
    import json
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    rows = []
    with torch.cuda.stream(st):
        for pname, shapes in PYRAMIDS.items():
            S = sum(h * w for h, w in shapes)
            for regime in ("decoder", "encoder"):
                Lq = 300 if regime == "decoder" else S
                for locs in ("uniform", "model"):
                    for N in (1, 2, 4, 8, 16, 32, 64):
                        WORKLOADS["_sweep"] = (N, shapes, 8, 32, Lq, 4)
                        _, d, dims = make_inputs("_sweep", 1000, dev, locs)
                        fb, bb = algorithmic_bytes(*dims)
                        table = _native.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64, with_table=True)[1]
                        fwd = lambda: _native.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64,
                                                                     with_table=True if table is not None else None)
                        bwd = lambda: _native.ms_deform_attn_backward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], d["go"], 64, table=table)
                        tf, tb = min(time_call(fwd, st), time_call(fwd, st)), min(time_call(bwd, st), time_call(bwd, st))   # (one-off hiccups of 20-300 us were seen on single measurements)
                        row = {"pyramid": pname, "regime": regime, "locations": locs, "N": N, "Lq": Lq, "S": S, "fwd_us": tf, "bwd_us": tb,
                               "samples_per_s": N / ((tf + tb) * 1e-6), "fwd_frac": fb / tf / 1e3 / 8000.0, "bwd_frac": bb / tb / 1e3 / 8000.0,
                               "plan": _native.describe_plan(N, S, 8, 32, len(shapes), Lq, 4), "table": table is not None}
                        rows.append(row)
                        print("%-6s %-7s %-7s N=%-3d fwd %8.2f us (%.3f)  bwd %8.2f us (%.3f)  %9.0f samples/s  %s" % (
                            pname, regime, locs, N, tf, row["fwd_frac"], tb, row["bwd_frac"], row["samples_per_s"], row["plan"]), flush=True)
                        del d, table
                        torch.cuda.empty_cache()
    cliffs = []
    for a, b in zip(rows, rows[1:]):
        if (a["pyramid"], a["regime"], a["locations"]) == (b["pyramid"], b["regime"], b["locations"]) and b["N"] == 2 * a["N"]:
            if b["samples_per_s"] < 0.85 * a["samples_per_s"]:
                cliffs.append({k: b[k] for k in ("pyramid", "regime", "locations", "N")} | {"drop": 1 - b["samples_per_s"] / a["samples_per_s"],
                                                                                             "plan_before": a["plan"], "plan_after": b["plan"]})
    with open(out_path, "w") as f:
        json.dump({"rows": rows, "cliffs": cliffs}, f, indent=1)
    print("cliffs (samples/s down > 15 %% when N doubles): %d" % len(cliffs))
    for c in cliffs:
        print("  ", c)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--sweep":
        return sweep(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "sweep.json"))
    names = sys.argv[1:] or list(WORKLOADS)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    for name in names:
        _, d, dims = make_inputs(name, 1000, dev, os.environ.get("KTIME_LOCATIONS", "uniform"))
        fb, bb = algorithmic_bytes(*dims)
        # KTIME_TABLE=1: the forward leaves its point table, the backward reads it (what the autograd Functions do)
        use_table = os.environ.get("KTIME_TABLE", "0") != "0"
        table = _native.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64, with_table=True)[1] if use_table else None
        fns = {"fwd": lambda: _native.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64,
                                                             with_table=True if use_table else None),
               "bwd": lambda: _native.ms_deform_attn_backward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], d["go"], 64,
                                                              deterministic=os.environ.get("KTIME_DETERMINISTIC", "0") != "0", table=table)}
        out = []
        with torch.cuda.stream(st):
            for k, fn in fns.items():
                fn(); st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for _ in range(10):
                        fn()
                for _ in range(3):
                    g.replay()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                for _ in range(20):
                    g.replay()
                b.record(st); b.synchronize()
                us = a.elapsed_time(b) * 1e3 / 200
                out.append("%s %8.2f us %6.0f GB/s" % (k, us, (fb if k == "fwd" else bb) / us / 1e3))
        print("%-13s %s   [%s]" % (name, "  ".join(out), " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith(("MSDA_", "KTIME_")))), flush=True)

if __name__ == "__main__":
    main()
