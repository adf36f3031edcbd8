MSDA_HEADMAJOR=1 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
for loc in uniform model; do for h in 0 1; do
KTIME_LOCATIONS=$loc MSDA_HEADMAJOR=$h timeout -k 10 200 python tools/ktime.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$loc hm=$h /"
done; done
