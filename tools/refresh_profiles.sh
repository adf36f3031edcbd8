#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: one bench line per workload and location
# distribution into gpurun_out/bench/ (copy the ones to be judged into profiles/).
set -u
OUT=gpurun_out/bench
mkdir -p $OUT
for w in cfg2_decoder cfg2_encoder cfg4_decoder cfg4_encoder; do
  timeout -k 10 300 python bench.py --workload $w > $OUT/${w}.json 2> $OUT/${w}.err && echo "$w done"
  timeout -k 10 300 python bench.py --workload $w --locations model --no-cpu-baseline > $OUT/${w}_model_locations.json 2>> $OUT/${w}.err && echo "$w model done"
done
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $OUT/cfg2_decoder_bf16.json 2>> $OUT/cfg2_decoder.err
timeout -k 10 300 python bench.py --graph-steps 1 --no-cpu-baseline > $OUT/cfg2_decoder_graph1.json 2>> $OUT/cfg2_decoder.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench/*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print("%-45s %10.0f samples/s  %.4f ms/step  fwd %.2f us  bwd %.2f us  frac %.3f  cpu %s" % (
        f.split("/")[-1], r["value"], r["ms_per_step"], r["kernels"]["fwd"]["ms"] * 1e3, r["kernels"]["bwd"]["ms"] * 1e3,
        r["roofline"]["frac"], ("%.1f" % r["cpu_baseline"]["value"]) if "cpu_baseline" in r else "-"))
PY
