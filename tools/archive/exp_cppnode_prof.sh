#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the 12-layer step with the attention modules as C++ nodes and as the Python
# composition; prints totals and the kernels whose totals differ most.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/cppnode
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in cpp py; do
  FLAG=""; [ $MODE = py ] && FLAG="--no-cpp-node"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$MODE -- python3 $GRAFT_REPO_ROOT/tools/ddp_step.py --steps 8 --warmup 2 $FLAG > $OUT/$MODE.log 2>&1
  cp $OUT/trace_$MODE/*/*_kernel_stats.csv $OUT/kernel_stats_$MODE.csv 2>/dev/null
  grep -o '"ms_per_step": [0-9.]*' $OUT/$MODE.log
done
python3 - $OUT <<'PY'
import csv, sys
def load(p):
    return {r["Name"][:100]: (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(p))}
a, b = load(sys.argv[1] + "/kernel_stats_cpp.csv"), load(sys.argv[1] + "/kernel_stats_py.csv")
print("total kernel time: C++ nodes %.1f ms, Python composition %.1f ms (10 steps each)" % (sum(v[0] for v in a.values()), sum(v[0] for v in b.values())))
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0,))[0] - b.get(k, (0,))[0]))
for k in keys[:16]:
    x, y = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    print("%-100s cpp %7.2f ms x%4d (%.1f us)   py %7.2f ms x%4d (%.1f us)" % (k, x[0], x[1], x[2], y[0], y[1], y[2]))
PY
