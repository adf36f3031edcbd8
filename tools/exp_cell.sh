#!/bin/bash
# cell-sorted role B: parity, timing (default vs deterministic), in-kernel stamps
set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_deterministic_gpu.py -x -q 2>&1 | tail -4 || exit 1
python tools/ktime.py 2>&1 | grep -v amdgpu.ids
KTIME_DETERMINISTIC=1 python tools/ktime.py 2>&1 | grep -v amdgpu.ids
KTIME_DETERMINISTIC=1 MSDA_CELL_SKIP_A=1 python tools/ktime.py 2>&1 | grep -v amdgpu.ids
for w in c2d c2e c4d c4e; do KB_DET=1 ./tools/micro/kbench $w 50 2>&1 | grep -E "cell role B|bwd:"; done
