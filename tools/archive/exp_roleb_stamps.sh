#!/bin/bash
# Phase breakdown (in-kernel stamps) of the default per-tap role B as its own kernel, all four workloads.
set -u
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS -DMSDA_TUNING -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o /tmp/kbench_x 2>&1 | grep error
for w in c2d c2e c4d c4e; do echo "== $w"; MSDA_BWD_MODE=split /tmp/kbench_x $w 50 2>&1 | grep -v "^cell"; done
