#!/bin/bash
# Instruction counts per wavefront of role A and role B of the small-problem backward, each alone in the fused launch, and of
# role B with phases switched off (stamped kbench build: KB_SKIP_ROLE, KB_DIAG): run ON THE GPU BOX from the repo root.
# usage: tools/pmc_roles.sh <outdir>
OUT=${1:-gpurun_out/roles}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp KB_TABLE=1
for v in "1 0" "2 0" "2 16" "2 19"; do
  set -- $v; export KB_SKIP_ROLE=$1 KB_DIAG=$2
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/skip$1_diag$2 -- $GRAFT_REPO_ROOT/tools/micro/kbench c2d 20 > $GRAFT_REPO_ROOT/$OUT/skip$1_diag$2.log 2>&1
done
python3 - $GRAFT_REPO_ROOT/$OUT <<'PY'
import csv, glob, sys, collections
for sk, dg, who in ((1, 0, "role A alone"), (2, 0, "role B alone"), (2, 16, "role B without its gather"), (2, 19, "role B without atomics, record writes, gather")):
    acc = collections.defaultdict(list)
    for p in glob.glob("%s/skip%d_diag%d/*/*_counter_collection.csv" % (sys.argv[1], sk, dg)):
        for r in csv.DictReader(open(p)):
            if "bwd_fused_d32" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    live = m["SQ_WAVES"] - (2048 if sk == 1 else 1200)          # the skipped role's waves retire after a handful of instructions
    print("%-50s" % who, {k.replace("SQ_INSTS_", ""): round(v / live, 1) for k, v in m.items() if k != "SQ_WAVES"})
PY
