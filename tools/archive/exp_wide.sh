#!/bin/bash
# kept-taps single pass (kAccWide) vs the chunked passes: role B alone (split) and the fused backward, stamps included.
set -u
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS -DMSDA_TUNING ${EXTRA:-} -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o /tmp/kbench_x 2>&1 | grep error
for w in ${WORKLOADS:-c2e c4e}; do
  kw=6; [ $w = c4e ] && kw=2     # ranges per level the plan picks (plan_value)
  for wide in 1 0; do
    echo "== $w wide=$wide fused"; MSDA_BWD_WIDE=$wide /tmp/kbench_x $w 50 2>&1 | grep -E "bwd:"
    echo "== $w wide=$wide split"
    if [ $wide = 1 ]; then KB_W=$kw KB_SKEW=0 MSDA_BWD_WIDE=1 MSDA_BWD_MODE=split /tmp/kbench_x $w 50 2>&1 | grep -E "bwd:|role B|loads|histogram|prefix|scatter|  gather|level" | head -12
    else MSDA_BWD_WIDE=0 MSDA_BWD_MODE=split /tmp/kbench_x $w 50 2>&1 | grep -E "bwd:|role B" | head -3; fi
  done
done
