import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench import make_inputs, WORKLOADS
from uvhand_amd import _native
import ktime
dev = torch.device("cuda", 0); st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    for pname, N, locs in (("p40x3", 4, "uniform"), ("p40x3", 2, "uniform"), ("p40x3", 8, "uniform"), ("p28", 2, "model"), ("p40x3", 4, "uniform")):
        shapes = ktime.PYRAMIDS[pname]
        WORKLOADS["_s"] = (N, shapes, 8, 32, 300, 4)
        _, d, dims = make_inputs("_s", 1000, dev, locs)
        for use_table in (True, False):
            table = _native.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64, with_table=True)[1] if use_table else None
            fwd = lambda: _native.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64, with_table=True if table is not None else None)
            bwd = lambda: _native.ms_deform_attn_backward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], d["go"], 64, table=table)
            print(pname, N, locs, "table" if table is not None else "scan", "fwd %.2f bwd %.2f" % (ktime.time_call(fwd, st), ktime.time_call(bwd, st)), flush=True)
