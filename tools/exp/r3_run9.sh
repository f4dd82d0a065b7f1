#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r9
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o pg -- python $GRAFT_REPO_ROOT/tools/exp/product_graph_trace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/pg -name "*kernel_trace.csv" | head -1)
python tools/kernel_gaps.py $f > gpurun_out/r9/product_graph_gaps.txt 2>&1
python - $f > gpurun_out/r9/product_graph_seq.txt <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last replay: from the last ref_draws_kernel on
idx = [i for i, r in enumerate(rows) if "ref_draws" in r[2]]
i0 = idx[-2]; i1 = idx[-1]
prev = rows[i0 - 1][1]
for s, e, n in rows[i0:i1]:
    print("%8.2f gap %6.2f dur %7.2f  %s" % ((s - rows[i0][0]) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n.split("(")[0][:90]))
    prev = e
print("replay period us", (rows[i1][0] - rows[i0][0]) / 1e3)
PY
