#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/cppnode
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in cpp py; do
  FLAG=""; [ $MODE = py ] && FLAG="--no-cpp-node"
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace_$MODE -- python3 $GRAFT_REPO_ROOT/tools/ddp_step.py --steps 6 --warmup 2 $FLAG > $OUT/k$MODE.log 2>&1
  f=$(ls $OUT/ktrace_$MODE/*/*_kernel_trace.csv | head -1)
  echo "== $MODE"; python3 $GRAFT_REPO_ROOT/tools/exp_gaps.py $f
  rm -rf $OUT/ktrace_$MODE
done
