#!/usr/bin/env python3
"""A/B of the LDS-stage kernels (msda_d32_lds.h) against the kernels they replace on large problems: bit equality of the
forward output and of grad_loc / grad_attn, and the per-call device times.  Diagnostic library (MSDA_LDS knob):
    python tools/exp_lds.py [workloads...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(mode, names):
    import torch
    from bench import make_inputs, WORKLOADS
    from uvhand_amd import _native
    _native.LIB_PATH = os.path.join(ROOT, "uvhand_amd", "libmsda_hip_tuning.so")
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    res = {}
    for name in names:
        for dt in os.environ.get("EXP_LDS_DTYPES", "f32 bf16").split():
            _, d, dims = make_inputs(name, 1000, dev, os.environ.get("KTIME_LOCATIONS", "uniform"))
            v = d["value"].to(torch.bfloat16) if dt == "bf16" else d["value"]
            go = d["go"].to(torch.bfloat16) if dt == "bf16" else d["go"]
            fwd = lambda: _native.ms_deform_attn_forward(v, d["shapes"], d["lsi"], d["loc"], d["attn"], 64)
            bwd = lambda: _native.ms_deform_attn_backward(v, d["shapes"], d["lsi"], d["loc"], d["attn"], go, 64,
                                                          fp32_grad_value=(dt == "bf16"))
            with torch.cuda.stream(st):
                out = fwd(); gv, gl, ga = bwd(); st.synchronize()
                times = []
                for fn in (fwd, bwd):
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
                    times.append(a.elapsed_time(b) * 1e3 / 200)
            torch.save({"out": out.cpu(), "gl": gl.cpu(), "ga": ga.cpu(), "gv": gv.float().cpu()}, "/tmp/lds_%s_%s_%s.pt" % (mode, name, dt))
            print("MSDA_LDS=%s %-13s %-4s fwd %8.2f us  bwd %8.2f us" % (mode, name, dt, times[0], times[1]), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3:])
        sys.exit(0)
    names = sys.argv[1:] or ["cfg2_encoder", "cfg4_decoder", "cfg4_encoder"]
    for mode in ("0", "1"):
        env = dict(os.environ, MSDA_LDS=mode)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", mode] + names, env=env)
    for wgs in os.environ.get("EXP_LDS_WGS", "").split():
        env = dict(os.environ, MSDA_LDS="1", MSDA_LDS_WGS=wgs)
        print("MSDA_LDS_WGS=%s" % wgs, flush=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", "w" + wgs] + names, env=env)
    import torch
    for name in names:
        for dt in os.environ.get("EXP_LDS_DTYPES", "f32 bf16").split():
            a, b = (torch.load("/tmp/lds_%s_%s_%s.pt" % (m, name, dt)) for m in ("0", "1"))
            print("%-13s %-4s out equal %s  grad_loc equal %s  grad_attn equal %s  grad_value max rel diff %.2e" % (
                name, dt, torch.equal(a["out"], b["out"]), torch.equal(a["gl"], b["gl"]), torch.equal(a["ga"], b["ga"]),
                float((a["gv"] - b["gv"]).abs().max() / a["gv"].abs().max())))
