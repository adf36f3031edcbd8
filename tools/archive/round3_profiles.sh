#!/bin/bash
# Runs ON THE GPU BOX (two gpurun calls: `tools/round3_profiles.sh a`, then `b`): everything profiles/r03_* is made from.
#   a: the bench line with its table, kernel times default / deterministic, weight-gradient times, f4 hygiene, host overhead
#   b: rocprofv3 kernel stats + the two PMC passes for the four workloads (fp32) and the headline in bf16; 12-layer step
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03
mkdir -p $OUT
if [ "${1:-a}" = "a" ]; then
  timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
  python tools/ktime.py > $OUT/ktime_default.log 2>&1
  KTIME_DETERMINISTIC=1 python tools/ktime.py > $OUT/ktime_deterministic.log 2>&1
  KTIME_LOCATIONS=model python tools/ktime.py > $OUT/ktime_model_locations.log 2>&1
  python tools/wgrad_time.py > $OUT/wgrad_time.log 2>&1; python tools/wgrad_time.py bf16 >> $OUT/wgrad_time.log 2>&1
  python tools/host_overhead.py > $OUT/host_overhead.log 2>&1
  bash tools/f4_hygiene.sh > $OUT/f4_hygiene.log 2>&1
  grep -hv amdgpu.ids $OUT/ktime_default.log $OUT/ktime_deterministic.log $OUT/ktime_model_locations.log $OUT/wgrad_time.log $OUT/host_overhead.log $OUT/f4_hygiene.log
else
  for w in cfg2_decoder cfg2_encoder cfg4_decoder cfg4_encoder; do bash tools/profile_gpu.sh r03 $w f32 > $OUT/profile_${w}_f32.log 2>&1; tail -4 $OUT/profile_${w}_f32.log; done
  bash tools/profile_gpu.sh r03 cfg2_decoder bf16 > $OUT/profile_cfg2_decoder_bf16.log 2>&1; tail -3 $OUT/profile_cfg2_decoder_bf16.log
  bash tools/layer_step.sh r03 > $OUT/layer_step.log 2>&1; grep -v amdgpu $OUT/layer_step.log | head -40
  python tools/ddp_step.py --steps 10 --warmup 3 --amp bf16 2>/dev/null | tee $OUT/layers_amp_bf16.json | cut -c1-200
fi
