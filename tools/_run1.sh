mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
bash tools/profile_gpu.sh r01y cfg2_decoder 2>&1 | tail -8
MODULE_GRAPH=1 timeout -k 10 120 python tools/module_step.py cfg2_decoder 50 2>&1 | tail -1
