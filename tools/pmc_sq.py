#!/usr/bin/env python3
"""SQ / TCP / TCC counters of the sampling kernels (run ON THE GPU BOX from the repo root):

    python3 tools/pmc_sq.py <tag> [workload ...]        -> gpurun_out/<tag>/<workload>_pmc_sq.json (+ the raw CSVs)

One `rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --workload <w> --no-graph ...` per counter set (the program goes
directly after `--`; no trace domain besides --kernel-trace; 8 SQ / 4 TCC slots per pass, MI355X_MICROARCH.md "rocprofv3 PMC
slots").  Counter names are checked against `rocprofv3 -L` first; names this ROCm does not know are dropped and listed.
Per kernel the JSON holds the mean of every counter over the kernel's dispatches plus a few derived ratios."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETS = [
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
     "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_INSTS_SMEM", "SQ_INSTS_FLAT",
     "SQ_INST_CYCLES_VMEM", "SQ_WAVE32_INSTS"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TOTAL_ACCESSES_sum"],
    ["TCP_TA_TCP_STATE_READ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_GATE_EN1_sum"],
    # (a TA_* set — TA_BUSY_avr, TA_TA_BUSY_sum, TA_ADDR_STALLED_BY_TC_CYCLES_sum, TA_DATA_STALLED_BY_TC_CYCLES_sum — hung rocprofv3 on
    # this pool for 7 minutes until the silence guard killed the call: not collected)
    ["SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_SMEM", "SQ_INSTS_LDS_ATOMIC",
     "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_WRITE_sum", "TCC_ATOMIC_sum"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
]


def available():
    try:
        out = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120).stdout
    except Exception as exc:                                   # no list: try every name, rocprofv3 rejects unknown ones per pass
        print("pmc_sq: rocprofv3 -L failed (%s)" % exc, file=sys.stderr)
        return None
    names = set()
    for tok in out.replace(",", " ").replace(":", " ").replace("|", " ").split():
        if tok[:3] in ("SQ_", "TCP", "TCC", "TA_", "GRB", "TD_", "SPI", "CPC"):
            names.add(tok.strip())
    return names


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "pmc"
    workloads = sys.argv[2:] or ["cfg2_decoder", "cfg2_encoder", "cfg4_decoder", "cfg4_encoder"]
    out_root = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out_root, exist_ok=True)
    avail = available()
    if avail is not None:
        with open(os.path.join(out_root, "counters_available.txt"), "w") as f:
            f.write("\n".join(sorted(avail)) + "\n")
    dropped = []
    env = dict(os.environ, TMPDIR="/tmp")
    for wl in workloads:
        per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
        only = [int(x) for x in os.environ.get("PMC_SETS", "").split(",") if x.strip()]
        for si, names in enumerate(SETS):
            if only and si not in only:
                continue
            use = [n for n in names if avail is None or n in avail]
            dropped += [n for n in names if n not in use]
            if not use:
                continue
            d = os.path.join(out_root, "raw_%s_set%d" % (wl, si))
            cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + use + ["--output-format", "csv", "-d", d, "--", "python3",
                   os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "30", "--warmup", "5", "--repeats", "1",
                   "--no-cpu-baseline", "--no-table", "--no-graph", "--kernel-iters", "10"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                print("pmc_sq: %s set %d TIMED OUT (%s)" % (wl, si, " ".join(use)), flush=True)
                continue
            print("pmc_sq: %s set %d rc=%d (%s)" % (wl, si, r.returncode, " ".join(use)), flush=True)
            if r.returncode != 0:
                print(r.stderr[-800:], file=sys.stderr)
                continue
            for path in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        k = row["Kernel_Name"].split("(")[0]
                        if "msda" in k:
                            per_kernel[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        res = {}
        for k, cs in per_kernel.items():
            m = {c: sum(v) / len(v) for c, v in cs.items()}
            m["dispatches"] = max(len(v) for v in cs.values())
            g = m.get
            if g("SQ_WAVES") and g("SQ_INSTS_VALU") is not None:
                m["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
                for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
                    if g(c) is not None:
                        m[c.lower().replace("sq_", "") + "_per_wave"] = m[c] / m["SQ_WAVES"]
            if g("SQ_WAVE_CYCLES"):
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                    if g(c) is not None:
                        m[c.lower().replace("sq_", "") + "_share_of_wave_cycles"] = m[c] / m["SQ_WAVE_CYCLES"]
            for lvl, cnt_name, out in (("SQ_IFETCH_LEVEL", "SQ_IFETCH", "ifetch_latency_cycles"), ("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM", "vmem_latency_cycles"),
                                       ("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS", "lds_latency_cycles"), ("SQ_INST_LEVEL_SMEM", "SQ_INSTS_SMEM", "smem_latency_cycles")):
                if g(lvl) is not None and g(cnt_name):
                    m[out] = m[lvl] / m[cnt_name]
            if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
                m["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
            if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
                m["l1_read_miss_per_access"] = m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"]
            res[k] = m
        with open(os.path.join(out_root, "%s_pmc_sq.json" % wl), "w") as f:
            json.dump({"workload": wl, "dropped_counters": sorted(set(dropped)), "kernels": res}, f, indent=1, sort_keys=True)
        for k, m in res.items():
            print("%s  %s" % (wl, k[:90]))
            for c in sorted(m):
                print("    %-44s %14.3f" % (c, m[c]))
    print("dropped (unknown to this rocprofv3):", sorted(set(dropped)))


if __name__ == "__main__":
    main()
