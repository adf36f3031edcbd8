#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times of role A and role B launched as separate kernels
# (diagnostic: MSDA_TUNING build knob MSDA_BWD_MODE=split), all workloads, uniform + model locations.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/split_${1:-r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MSDA_BWD_MODE=split
for LOC in uniform model; do
  export KTIME_LOCATIONS=$LOC
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$LOC -- python3 $GRAFT_REPO_ROOT/tools/ktime.py > $OUT/ktime_$LOC.log 2>&1
  cp $OUT/trace_$LOC/*/*_kernel_stats.csv $OUT/kernel_stats_$LOC.csv 2>/dev/null
  cat $OUT/ktime_$LOC.log | grep -v amdgpu.ids
done
