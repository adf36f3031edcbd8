#!/bin/bash
# per-level phase times of role B at cfg-4 decoder (single pass, W = 1) and cfg-2 decoder
set -u
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS -DMSDA_TUNING ${EXTRA:-} -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o /tmp/kbench_x 2>&1 | grep error
echo "== c4d fused"; /tmp/kbench_x c4d 50 2>&1 | grep -E "bwd:"
echo "== c4d split"; KB_W=1 KB_SKEW=1 MSDA_BWD_MODE=split /tmp/kbench_x c4d 50 2>&1 | grep -E "bwd:|role B|loads|histogram|prefix|scatter|  gather|level" | head -12
echo "== c2d fused"; /tmp/kbench_x c2d 200 2>&1 | grep -E "bwd:"
