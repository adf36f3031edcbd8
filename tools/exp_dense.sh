#!/bin/bash
# usage (on the GPU box): tools/exp_dense.sh [workloads] — role B alone and the whole backward, with the dense coarse-level path
# (default) and without it (KB_DIAG=128), plus role B's per-level phase stamps (KB_W = ranges per level of the plan)
set -u
cd $GRAFT_REPO_ROOT
for wl in ${@:-c2e c4e c4d}; do
  W=$(case $wl in c2e) echo 6;; c4e) echo 2;; *) echo 1;; esac)
  echo "=== $wl (W = $W)"
  for d in 0 128; do
    echo -n "role B alone, diag $d: "; KB_SKIP_ROLE=2 KB_DIAG=$d tools/exp_kbench.sh $wl | grep " bwd:"
    echo -n "both roles,   diag $d: "; KB_DIAG=$d tools/exp_kbench.sh $wl | grep " bwd:"
    KB_DIAG=$d KB_W=$W KB_SKEW=0 tools/exp_kbench.sh $wl | grep "level "
  done
done
