cd $GRAFT_REPO_ROOT
export KTIME_TABLE=1
for w in 256 320 384 448 512 640; do MSDA_BWD_WGS=$w python tools/ktime.py cfg2_encoder 2>/dev/null; done
for a in 128 192 256 384 512; do MSDA_LDS_WGS_BWD=$a python tools/ktime.py cfg2_encoder 2>/dev/null; done
for a in 256 512; do MSDA_BWD_WGS=512 MSDA_LDS_WGS_BWD=$a python tools/ktime.py cfg2_encoder 2>/dev/null; done
for w in 512 1024; do MSDA_BWD_WGS=$w python tools/ktime.py cfg4_encoder 2>/dev/null; done
for a in 512 2048; do MSDA_LDS_WGS_BWD=$a python tools/ktime.py cfg4_encoder 2>/dev/null; done
