timeout -k 10 300 python -m pytest tests/test_linear_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 100 python tools/wgrad_time.py 2>&1 | grep -v amdgpu.ids
