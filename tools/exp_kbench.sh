#!/bin/bash
# usage (on the GPU box): [ENV...] tools/exp_kbench.sh <workload> [KB_W]   — stamped unity build, per-phase times of role B
set -u
cd $GRAFT_REPO_ROOT
if [ ! -x /tmp/kbench_y ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS -DMSDA_TUNING ${EXTRA:-} -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o /tmp/kbench_y 2>&1 | grep error
fi
/tmp/kbench_y $1 50 2>&1 | grep -v "^$"
