#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of msda_linear_forward_f32 and torch's addmm at rows = 33440, out = in = 256.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc
mkdir -p $OUT
cat > /tmp/gemm_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from uvhand_amd import _native
rows, out_f, in_f = 33440, 256, 256
x, w, b = torch.randn(rows, in_f, device="cuda"), torch.randn(out_f, in_f, device="cuda") * 0.1, torch.randn(out_f, device="cuda")
for _ in range(20):
    _native.linear_forward(x, w, b)
    torch.addmm(b, x, w.t())
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 /tmp/gemm_one.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- python3 /tmp/gemm_one.py > $OUT/p2.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for p in ("p1", "p2"):
    fs = glob.glob(sys.argv[1] + "/" + p + "/*/*counter_collection.csv")
    if not fs:
        print(p, "no counters; log tail:"); print(open(sys.argv[1] + "/" + p + ".log").read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        if "linear_rows" in k or "Cijk" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print("   %-32s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
