#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of dense_stream_kernel per ablation value
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06h}
shift
mkdir -p $OUT
for d in "$@"; do
  rocprofv3 --kernel-trace --stats -d $OUT/k$d -o k -- python tools/exp/r6_dense_abl.py $d > /dev/null 2> $OUT/k$d.err
  echo "dbg $d: $(python tools/rocpd_stats.py $OUT/k$d/k_results.db | grep -E 'dense_stream|dense_stats' | awk '{print $1, $3}' | tr '\n' ' ')" | tee -a $OUT/abl.txt
  rm -rf $OUT/k$d
done
