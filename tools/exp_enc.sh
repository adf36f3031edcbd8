#!/bin/bash
# usage (on the GPU box): tools/exp_enc.sh [workloads] — encoder-regime backward: the plan, roles alone and role B's phases (stamped unity build)
set -u
cd $GRAFT_REPO_ROOT
for wl in ${@:-c2e c4e c4d}; do
  echo "=== $wl"
  python - <<P
import uvhand_amd._native as n
N, S, Lq = {'c2e': (2, 3060, 3060), 'c4e': (32, 1045, 1045), 'c4d': (32, 1045, 300), 'c2d': (2, 3060, 300)}['$wl']
print(n.describe_plan(N, S, 8, 32, 4, Lq, 4))
P
  tools/exp_kbench.sh $wl
  KB_SKIP_ROLE=1 tools/exp_kbench.sh $wl | grep "bwd\|skipping"
  KB_SKIP_ROLE=2 tools/exp_kbench.sh $wl | grep "bwd\|skipping"
done
