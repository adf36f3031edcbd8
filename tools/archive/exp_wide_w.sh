#!/bin/bash
# kAccWide: ranges per level (MSDA_WIDE_WGS x 256 workgroups aimed for) vs backward time
set -u
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS -DMSDA_TUNING ${EXTRA:-} -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o /tmp/kbench_y 2>&1 | grep error
for k in 2 3 4 6; do echo "c2e MSDA_WIDE_WGS=$k"; MSDA_WIDE_WGS=$k /tmp/kbench_y c2e 100 2>&1 | grep -E "bwd:"; done
for k in 2 16 24 32; do echo "c4e MSDA_WIDE_WGS=$k"; MSDA_WIDE_WGS=$k /tmp/kbench_y c4e 100 2>&1 | grep -E "bwd:"; done
