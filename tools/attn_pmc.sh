#!/bin/bash
# Runs ON THE GPU BOX from the repo root: rocprofv3 kernel stats and SQ counters of the attention-core kernels (tools/attn_time.py).
#   gpurun_out/attn_<tag>/kernel_stats.csv, pmc_<set>.csv, summary.json
# Counter passes carry no trace domain besides --kernel-trace; the program goes directly after `--`.
set -u
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/attn_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/attn_time.py > $OUT/trace.log 2>&1
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
export ATTN_EAGER=1
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_$i -- python3 $GRAFT_REPO_ROOT/tools/attn_time.py > $OUT/pmc_$i.log 2>&1
  cp $OUT/pmc_$i/*/*_counter_collection.csv $OUT/pmc_$i.csv 2>/dev/null
done
python3 - "$OUT" <<'PY'
import csv, json, sys, os, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for i in (1, 2):
    path = os.path.join(out, "pmc_%d.csv" % i)
    if not os.path.exists(path):
        print("missing", path); continue
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        if "attn32" in k:
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in acc.items():
        res[k][c] = sum(v) / len(v)
for k, d in res.items():
    if d.get("SQ_BUSY_CYCLES") and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        d["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
    if d.get("SQ_WAVES"):
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
            if c in d: d[c + "_per_wave"] = d[c] / d["SQ_WAVES"]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    print(k, json.dumps({a: round(b, 3) for a, b in d.items()}))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(open(os.path.join(out, "kernel_stats.csv")).read()[:900])
PY
