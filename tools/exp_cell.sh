#!/bin/bash
# experiment: role B alone / role A alone on the cell path, 512- vs 256-thread workgroups
set -u
cd $GRAFT_REPO_ROOT
echo "== 512 threads: full / B only / A only"
python tools/ktime.py 2>&1 | grep -v amdgpu.ids
MSDA_CELL_SKIP_A=1 python tools/ktime.py 2>&1 | grep -v amdgpu.ids
MSDA_CELL_SKIP_B=1 python tools/ktime.py 2>&1 | grep -v amdgpu.ids
echo "== 256 threads"
make -s -C uvhand_amd/csrc clean && make -s -C uvhand_amd/csrc EXTRA=-DMSDA_CELL_THREADS=256 2>&1 | grep -E "error" 
python tools/ktime.py 2>&1 | grep -v amdgpu.ids
MSDA_CELL_SKIP_A=1 python tools/ktime.py 2>&1 | grep -v amdgpu.ids
MSDA_CELL_SKIP_B=1 python tools/ktime.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
