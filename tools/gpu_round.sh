#!/bin/bash
# usage: tools_gpu_round.sh  (runs on the GPU box from repo root)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for w in cfg2_decoder cfg2_encoder cfg4_decoder cfg4_encoder; do
  timeout -k 10 200 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        r=json.loads(line); print('$w', 'samples/s %.0f'%r['value'], 'ms/step %.4f'%r['ms_per_step'], 'fwd us %.2f (%.0f GB/s)'%(r['kernels']['fwd']['ms']*1e3, r['kernels']['fwd']['GBps']), 'bwd us %.2f (%.0f GB/s)'%(r['kernels']['bwd']['ms']*1e3, r['kernels']['bwd']['GBps']))
"
done
