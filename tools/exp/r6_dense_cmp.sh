#!/bin/bash
# round 6: dense correspondence - variants by STEGO_DEBUG (0: prep + dense_dma_kernel; 262144: stats + dense_stream_kernel; 2097152: prep + row-block kernel): tests, kernel times
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06i}
shift
mkdir -p $OUT
for dbg in "$@"; do
  STEGO_DEBUG=$dbg timeout 600 python -m pytest tests/test_dense_corr.py -x -q -m gpu 2>&1 | tail -1
  STEGO_DEBUG=$dbg rocprofv3 --kernel-trace --stats -d $OUT/kd$dbg -o kd -- python tools/bench_dense.py > $OUT/dense_$dbg.json 2> $OUT/kd.err
  cut -c1-230 $OUT/dense_$dbg.json
  python tools/rocpd_stats.py $OUT/kd$dbg/kd_results.db | grep -i "dense" | grep -v tile_kernel | cut -c1-120
  rm -rf $OUT/kd$dbg
done
