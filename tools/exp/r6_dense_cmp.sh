#!/bin/bash
# round 6: dense correspondence - the streaming kernel (default) against the row-block kernel with the coalesced prologue (STEGO_DEBUG bit 21): tests of both, kernel times
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06i}
mkdir -p $OUT
for dbg in 0 2097152; do
  STEGO_DEBUG=$dbg timeout 600 python -m pytest tests/test_dense_corr.py -x -q -m gpu 2>&1 | tail -1
  STEGO_DEBUG=$dbg rocprofv3 --kernel-trace --stats -d $OUT/kd$dbg -o kd -- python tools/bench_dense.py > $OUT/dense_$dbg.json 2> $OUT/kd.err
  cut -c1-230 $OUT/dense_$dbg.json
  python tools/rocpd_stats.py $OUT/kd$dbg/kd_results.db | grep -i "dense" | cut -c1-120
  rm -rf $OUT/kd$dbg
done
