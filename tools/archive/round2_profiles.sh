#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/r02_* is made from.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02
mkdir -p $OUT
for w in cfg2_decoder cfg2_encoder cfg4_decoder cfg4_encoder; do
  timeout -k 10 300 python bench.py --workload $w > $OUT/bench_${w}.json 2> $OUT/bench_${w}.err && echo "$w done"
done
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_cfg2_decoder_bf16.json 2>> $OUT/bench_cfg2_decoder.err
timeout -k 10 300 python bench.py --locations model --no-cpu-baseline > $OUT/bench_cfg2_decoder_model_locations.json 2>> $OUT/bench_cfg2_decoder.err
python tools/ktime.py > $OUT/ktime_default.log 2>&1
KTIME_DETERMINISTIC=1 python tools/ktime.py > $OUT/ktime_deterministic.log 2>&1
python tools/host_overhead.py > $OUT/host_overhead.log 2>&1
./tools/micro/gather_width > $OUT/gather_width.log 2>&1
for w in cfg2_decoder cfg2_encoder cfg4_decoder cfg4_encoder; do MODULE_GRAPH=1 python tools/module_step.py $w; done > $OUT/module_step.log 2>&1
bash tools/profile_gpu.sh r02 cfg2_decoder f32 > $OUT/profile_cfg2_decoder_f32.log 2>&1
bash tools/profile_gpu.sh r02 cfg2_decoder bf16 > $OUT/profile_cfg2_decoder_bf16.log 2>&1
grep -v amdgpu.ids $OUT/ktime_default.log; grep -v amdgpu.ids $OUT/ktime_deterministic.log; cat $OUT/host_overhead.log | grep -v amdgpu; cat $OUT/module_step.log | grep -v amdgpu
