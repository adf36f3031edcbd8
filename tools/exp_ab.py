#!/usr/bin/env python3
"""A/B of diagnostic-library knob settings on the op's forward / backward: per-call device times (HIP graph of 10 calls) and
how far every result is from the FIRST setting's (bit-equal or max difference relative to the tensor's max).
    EXP_SETS="MSDA_LDS=0;MSDA_LDS=1 MSDA_LM=0;MSDA_LDS=1" python tools/exp_ab.py [workloads...]
    EXP_DTYPES="f32 bf16"   KTIME_LOCATIONS=model"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DTYPES = os.environ.get("EXP_DTYPES", "f32").split()

def child(tag, names):
    import torch
    from bench import make_inputs
    from uvhand_amd import _native
    _native.LIB_PATH = os.path.join(ROOT, "uvhand_amd", "libmsda_hip_tuning.so")
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    for name in names:
        for dt in DTYPES:
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
            torch.save({"out": out.float().cpu(), "gl": gl.cpu(), "ga": ga.cpu(), "gv": gv.float().cpu()},
                       "/tmp/ab_%s_%s_%s.pt" % (tag, name, dt))
            print("[%s] %-13s %-4s fwd %8.2f us  bwd %8.2f us" % (os.environ.get("EXP_SET", ""), name, dt, times[0], times[1]), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3:])
        sys.exit(0)
    names = sys.argv[1:] or ["cfg2_encoder", "cfg4_decoder", "cfg4_encoder"]
    sets = [x.strip() for x in os.environ.get("EXP_SETS", "MSDA_LDS=0;MSDA_LDS=1").split(";") if x.strip()]
    for i, st in enumerate(sets):
        env = dict(os.environ, EXP_SET=st)
        env.update(dict(kv.split("=", 1) for kv in st.split()))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", str(i)] + names, env=env)
    import torch
    def cmp(a, b):
        return "equal" if torch.equal(a, b) else "%.1e" % float((a - b).abs().max() / a.abs().max())
    for name in names:
        for dt in DTYPES:
            ref = torch.load("/tmp/ab_0_%s_%s.pt" % (name, dt))
            for i, st in enumerate(sets[1:], 1):
                t = torch.load("/tmp/ab_%d_%s_%s.pt" % (i, name, dt))
                print("%-13s %-4s [%s] vs [%s]: out %s  grad_loc %s  grad_attn %s  grad_value %s" % (
                    name, dt, st, sets[0], cmp(ref["out"], t["out"]), cmp(ref["gl"], t["gl"]), cmp(ref["ga"], t["ga"]), cmp(ref["gv"], t["gv"])))
