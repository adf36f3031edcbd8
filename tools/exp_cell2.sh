#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for P in 0 512 1024; do
  echo "== MSDA_CELL_PERSIST=$P"
  MSDA_CELL_PERSIST=$P KTIME_DETERMINISTIC=1 MSDA_CELL_SKIP_A=1 python tools/ktime.py cfg2_encoder cfg4_decoder cfg4_encoder 2>&1 | grep -v amdgpu.ids
done
