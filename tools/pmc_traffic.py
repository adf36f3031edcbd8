#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC summaries tools/profile_gpu.sh leaves under gpurun_out/prof_<tag>_<workload>_<dtype>/.

    python tools/pmc_traffic.py r03            # after `bash tools/profile_gpu.sh r03 <workload> <dtype>` for each workload

HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports KiB, and on gfx950 FETCH_SIZE counts
half the bytes of a wide coalesced read (MI355X_MICROARCH.md, section HBM; calibrated for this project's 8-lane x 16-B row
reads in profiles/r01_notes.md).  The table is stamped with the fingerprint of the kernel sources (bench.kernel_sources_sha16):
bench.py reports `roofline.traffic` only while the kernels are the ones that were profiled.  Also copies each summary and
kernel-stats file to profiles/<tag>_bench_<workload>[_bf16]_{pmc_summary.json,kernel_stats.csv}."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, kernel_sources_sha16          # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    table = {"_doc": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (rocprofv3 --pmc, separate passes; "
                     "FETCH_SIZE doubled on gfx950 per MI355X_MICROARCH.md, calibrated for this access shape in "
                     "profiles/r01_notes.md).  Written by tools/pmc_traffic.py from profiles/%s_bench_<workload>[_bf16]_pmc_summary.json; "
                     "valid for the kernel sources with fingerprint sources_sha16 only." % tag,
             "sources_sha16": kernel_sources_sha16()}
    for wl in WORKLOADS:
        for dt in ("f32", "bf16"):
            d = os.path.join(ROOT, "gpurun_out", "prof_%s_%s_%s" % (tag, wl, dt))
            path = os.path.join(d, "pmc_summary.json")
            if not os.path.exists(path):
                continue
            with open(path) as f:
                summ = json.load(f)
            key = wl + ("_bf16" if dt == "bf16" else "")
            stem = os.path.join(ROOT, "profiles", "%s_bench_%s" % (tag, key))
            shutil.copyfile(path, stem + "_pmc_summary.json")
            if os.path.exists(os.path.join(d, "kernel_stats.csv")):
                shutil.copyfile(os.path.join(d, "kernel_stats.csv"), stem + "_kernel_stats.csv")
            entry = {}
            for kernel, c in summ.items():
                if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                    continue
                name = kernel.replace("void ", "")
                which = "fwd" if "fwd_" in name else "bwd" if ("bwd_fused" in name or "bwd_gather" in name) else None
                if which is None:
                    which = name                                  # helper kernels keep their own name
                entry[which] = {"kernel": name, "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
                                "hbm_bytes_per_launch": int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))}
            table[key] = entry
            print(key, {k: v["hbm_bytes_per_launch"] for k, v in entry.items()})
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(table, f, indent=1)
    print("profiles/pmc_traffic.json written for sources", table["sources_sha16"])


if __name__ == "__main__":
    main()
