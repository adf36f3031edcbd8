mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 200 python tools/ktime.py 2>&1 | grep -v amdgpu.ids
KTIME_LOCATIONS=model timeout -k 10 200 python tools/ktime.py 2>&1 | grep -v amdgpu.ids
