#!/usr/bin/env python3
"""Idle gaps on the GPU from a rocprofv3 --kernel-trace CSV: total busy / idle time and the kernels that follow the largest gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows), key=lambda e: e[0])
skip = int(len(ev) * 0.25)                      # warm-up steps
ev = ev[skip:]
busy = sum(e[1] - e[0] for e in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.Counter(); cnt = collections.Counter()
last_end = ev[0][1]
for s, e, n in ev[1:]:
    g = s - last_end
    if g > 5000:
        gaps[n] += g; cnt[n] += 1
    last_end = max(last_end, e)
print("span %.1f ms, busy %.1f ms, idle %.1f ms over %d dispatches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(ev)))
for n, g in gaps.most_common(12):
    print("  %8.2f ms idle before %4d x %s" % (g / 1e6, cnt[n], n))
