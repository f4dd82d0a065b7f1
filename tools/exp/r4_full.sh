#!/bin/bash
# round 4: the whole GPU suite + the kernels of one replayed product step
export TMPDIR=/tmp
O=gpurun_out/r4_full
mkdir -p $O
(rocm-smi --showclocks --showperflevel; rocminfo | grep -i -E "compute unit|partition" | head -8) > $O/box.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $O/tests.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o pg -- python $GRAFT_REPO_ROOT/tools/exp/product_graph_trace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/pg -name "*kernel_trace.csv" | head -1)
python - $f > $O/product_graph_seq.txt <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "ref_draws" in r[2]]
i0 = idx[-2]; i1 = idx[-1]
prev = rows[i0 - 1][1]; t0 = rows[i0][0]
for s, e, n in rows[i0:i1]:
    print("%8.2f gap %6.2f dur %7.2f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n[:100]))
    prev = e
print("replay period us", (rows[i1][0] - rows[i0][0]) / 1e3)
PY
cat $O/tests.txt $O/product_graph_seq.txt
