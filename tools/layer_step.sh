#!/bin/bash
# Runs ON THE GPU BOX: the 12-layer train-step harness (tools/ddp_step.py, window 32) with the drop-in layers' fused
# add+LayerNorm / MFMA FFN weight gradients ("fused") and with stock PyTorch ops in their place ("plain"), each
# under rocprofv3 --kernel-trace --stats.  Output: gpurun_out/layers_<tag>/{fused,plain}.json + kernel_stats_*.csv
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/layers_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in fused plain; do
  FLAG=""; [ $MODE = plain ] && FLAG="--plain-layers"
  python3 $GRAFT_REPO_ROOT/tools/ddp_step.py --steps 10 --warmup 3 $FLAG > $OUT/$MODE.json 2> $OUT/$MODE.err
  cat $OUT/$MODE.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$MODE -- python3 $GRAFT_REPO_ROOT/tools/ddp_step.py --steps 5 --warmup 2 $FLAG > $OUT/${MODE}_trace.log 2>&1
  cp $OUT/trace_$MODE/*/*_kernel_stats.csv $OUT/kernel_stats_$MODE.csv 2>/dev/null
  python3 - "$OUT/kernel_stats_$MODE.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("  total kernel time %.2f ms over %d kernels" % (tot / 1e6, len(rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("  %6.2f%% %9.1f us x %5s  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:110]))
PY
done
