#!/bin/bash
# cell kernel occupancy experiment: tile rows / pending capacity / min waves per SIMD (diagnostic builds)
set -u
cd $GRAFT_REPO_ROOT
run() {
  echo "== EXTRA=$1"
  make -s -C uvhand_amd/csrc clean >/dev/null; make -s -C uvhand_amd/csrc tuning EXTRA="$1" 2>&1 | grep -E "error"
  KTIME_DETERMINISTIC=1 MSDA_CELL_SKIP_A=1 python tools/ktime.py 2>&1 | grep -v amdgpu.ids
}
run ""
run "-DMSDA_CELL_TILE_ROWS=128 -DMSDA_CELL_MIN_WAVES=6"
run "-DMSDA_CELL_TILE_ROWS=128 -DMSDA_CELL_PEND=512 -DMSDA_CELL_MIN_WAVES=8"
run "-DMSDA_CELL_TILE_ROWS=128 -DMSDA_CELL_PEND=512 -DMSDA_CELL_MIN_WAVES=4"
make -s -C uvhand_amd/csrc clean >/dev/null
