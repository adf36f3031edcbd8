#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
run() {
  echo "== EXTRA=$1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS -DMSDA_TUNING $1 -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o /tmp/kbench_x 2>&1 | grep error
  for w in c2d c2e c4d c4e; do KB_DET=1 MSDA_CELL_SKIP_A=1 /tmp/kbench_x $w 30 2>&1 | grep -E "bwd:|cell role B|per CU"; done
}
run ""
run "-DMSDA_CELL_T4=1 -DMSDA_CELL_TILE_ROWS=64"
