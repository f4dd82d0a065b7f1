#!/usr/bin/env python
"""From a rocprofv3 kernel trace CSV: per-kernel mean duration and the mean gap to the previous kernel, over the steady-state steps."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[len(rows) // 2:]           # steady state: the second half
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for s, e, n in rows:
    n = n.split("(")[0][:70]
    dur[n].append(e - s)
    if prev_end is not None: gap[n].append(s - prev_end)
    prev_end = e
print("%-72s %8s %10s %10s" % ("kernel", "calls", "dur_us", "gap_before_us"))
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    g = gap[n]
    print("%-72s %8d %10.2f %10.2f" % (n, len(dur[n]), sum(dur[n]) / len(dur[n]) / 1e3, (sorted(g)[len(g) // 2] / 1e3) if g else 0))
