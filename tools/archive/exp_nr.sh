#!/bin/bash
# gather_rows on the fixed-capacity path, one lane group per row: 4 rows x 2 records in flight (default) vs 2 rows x 4
set -u
cd $GRAFT_REPO_ROOT
for nr in 4 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS -DMSDA_TUNING -DMSDA_GATHER_NR1=$nr -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o /tmp/kbench_n$nr 2>&1 | grep error
done
for r in 1 2 3; do for nr in 4 2; do echo "NR1=$nr"; /tmp/kbench_n$nr c2d 500 2>&1 | grep -E "bwd:"; /tmp/kbench_n$nr c4d 200 2>&1 | grep -E "bwd:"; done; done
