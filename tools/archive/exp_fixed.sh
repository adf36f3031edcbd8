#!/bin/bash
# cfg-2 decoder: fixed-capacity sort (default) vs prefix-sum sort + balanced gather
set -u
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS -DMSDA_TUNING ${EXTRA:-} -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o /tmp/kbench_y 2>&1 | grep error
for f in 1 0; do for r in 1 2 3; do echo "MSDA_BWD_FIXED=$f"; MSDA_BWD_FIXED=$f /tmp/kbench_y c2d 500 2>&1 | grep -E "bwd:"; done; done
MSDA_BWD_FIXED=0 KB_W=4 MSDA_BWD_MODE=split /tmp/kbench_y c2d 200 2>&1 | grep -E "role B|loads|histogram|prefix|scatter|  gather|level" | head -12
