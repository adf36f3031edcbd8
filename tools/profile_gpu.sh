#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root.  Produces, under gpurun_out/prof_<tag>/:
#   kernel_stats.csv       rocprofv3 --kernel-trace --stats of `bench.py` (default workload)
#   pmc_fetch.csv / pmc_write.csv   per-dispatch FETCH_SIZE / WRITE_SIZE (separate passes)
# and prints the per-kernel averages.  Counters are collected WITHOUT any trace domain besides
# --kernel-trace (gpurun refuses --pmc combined with sys/hip/hsa traces).
set -u
TAG=${1:-r02}
WL=${2:-cfg2_decoder}
DT=${3:-f32}                      # f32 | bf16
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${WL}_${DT}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --dtype $DT --steps 300 --warmup 30 --repeats 5 --no-cpu-baseline --no-table > $OUT/bench_trace.log 2>&1
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --dtype $DT --steps 50 --warmup 5 --repeats 1 --no-cpu-baseline --no-table --no-graph > $OUT/bench_pmc_$C.log 2>&1
  cp $OUT/pmc_$C/*/*_counter_collection.csv $OUT/pmc_$C.csv 2>/dev/null
done
python3 - "$OUT" "$WL" <<'PY'
import csv, json, sys, collections, os
out, wl = sys.argv[1], sys.argv[2]
print(open(os.path.join(out, "kernel_stats.csv")).read()[:1500])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(out, "pmc_%s.csv" % c)
    if not os.path.exists(path):
        print("missing", path); continue
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == c:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        if "msda" in k:
            res.setdefault(k, {})[c] = sum(v) / len(v)
            print("%-60s %s mean %.1f KiB over %d dispatches" % (k[:60], c, sum(v) / len(v), len(v)))
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
PY
