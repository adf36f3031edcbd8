# dense-A (role A's coarse levels on the matrix cores): A/B per shape
cd $GRAFT_REPO_ROOT
export KTIME_TABLE=1
for wl in cfg2_encoder cfg4_decoder cfg4_encoder; do
  for a in 0 1 0 1; do MSDA_DENSE_A=$a python tools/ktime.py $wl 2>/dev/null; done
done
