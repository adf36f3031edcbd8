#!/bin/bash
# Runs ON THE GPU BOX from the repo root: the GPU suite, then the bench line (headline + table) and the four workloads' kernel times.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"
python - <<'PY'
import json
try:
    r = json.loads([l for l in open("gpurun_out/bench_default.json") if l.startswith("{")][-1])
except Exception as e:
    print("bench line unreadable:", e); raise SystemExit(0)
print("headline %.0f samples/s  %.4f ms/step  fwd %.2f us  bwd %.2f us  frac %.3f  eager %.1f us  cpu %s" % (
    r["value"], r["ms_per_step"], r["kernels"]["fwd"]["ms"] * 1e3, r["kernels"]["bwd"]["ms"] * 1e3, r["roofline"]["frac"],
    1e3 * r.get("eager_ms_per_step", 0), r.get("cpu_baseline", {}).get("value")))
for w in r.get("workloads", []):
    if "error" in w: print(w); continue
    print("%-13s %-4s %-7s %-3s step %8.2f us  fwd %7.2f (%.3f)  bwd %7.2f (%.3f)  %9.0f samples/s" % (
        w["workload"], w["dtype"], w.get("locations", "uniform"), "det" if w.get("deterministic") else "", w["step_us"], w["fwd_us"],
        w["fwd_frac"], w["bwd_us"], w["bwd_frac"], w["samples_per_s"]))
for m in r.get("modules", []):
    print("module", m)
for m in r.get("layers", []):
    print("layer", m)
PY
