#!/bin/bash
# usage (on the GPU box): tools/exp_enc_diag.sh [workloads] — role B alone (KB_SKIP_ROLE=2) with its phases switched off one by one
# (timing only; KB_DIAG bits: 4 = 2^2 no row loads in the gather, 16 = no gather, 32 = no scatter + gather, 64 = scan only)
set -u
cd $GRAFT_REPO_ROOT
for wl in ${@:-c2e c4e c4d}; do
  echo "=== $wl"
  for d in 0 4 16 32 64; do
    echo -n "role B alone, diag $d: "; KB_SKIP_ROLE=2 KB_DIAG=$d tools/exp_kbench.sh $wl | grep " bwd:"
  done
  echo -n "role A alone: "; KB_SKIP_ROLE=1 tools/exp_kbench.sh $wl | grep " bwd:"
  echo -n "both: "; tools/exp_kbench.sh $wl | grep " bwd:"
done
