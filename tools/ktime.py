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

def main():
    names = sys.argv[1:] or list(WORKLOADS)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    for name in names:
        _, d, dims = make_inputs(name, 1000, dev, os.environ.get("KTIME_LOCATIONS", "uniform"))
        fb, bb = algorithmic_bytes(*dims)
        fns = {"fwd": lambda: _native.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64),
               "bwd": lambda: _native.ms_deform_attn_backward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], d["go"], 64,
                                                              deterministic=os.environ.get("KTIME_DETERMINISTIC", "0") != "0")}
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
        print("%-13s %s   [%s]" % (name, "  ".join(out), " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("MSDA_"))), flush=True)

if __name__ == "__main__":
    main()
