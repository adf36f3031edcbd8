#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of one MSDeformAttn MODULE forward+backward loop (the op plus its four
# projections, fused prologue, weight-gradient kernels), fp32 and bf16 rows, two shapes.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/module_${1:-r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in cfg2_decoder cfg4_encoder; do
  for amp in "" bf16; do
    tag=${w}${amp:+_$amp}
    MODULE_AMP=$amp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $GRAFT_REPO_ROOT/tools/module_step.py $w 30 > $OUT/$tag.log 2>&1
    cp $OUT/trace_$tag/*/*_kernel_stats.csv $OUT/kernel_stats_$tag.csv 2>/dev/null
    grep -v "amdgpu\|rocprofv3\|^W2\|^E2" $OUT/$tag.log | tail -2
    python3 - "$OUT/kernel_stats_$tag.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = max(int(r["Calls"]) for r in rows if "msda::fwd_d32" in r["Name"])
print("  kernel time per step %.1f us over %d steps" % (tot / calls / 1e3, calls))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("  %6.2f%% %8.1f us x %4s  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:100]))
PY
  done
done
