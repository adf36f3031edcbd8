#!/bin/bash
# fused backward + forward, all four workloads (kbench, eager launches, HIP events)
set -u
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS -DMSDA_TUNING ${EXTRA:-} -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o /tmp/kbench_y 2>&1 | grep error
for w in c2d c2e c4d c4e; do /tmp/kbench_y $w 100 2>&1 | grep -E "fwd:|bwd:"; done
