#!/bin/bash
# round 6: per-kernel times of the dense correspondence path (tools/bench_dense.py under rocprofv3 --kernel-trace --stats)
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06f}
mkdir -p $OUT
python tools/bench_dense.py > $OUT/dense.json 2> $OUT/dense.err; cat $OUT/dense.json | cut -c1-600
rocprofv3 --kernel-trace --stats -d $OUT/kd -o kd -- python tools/bench_dense.py > /dev/null 2> $OUT/kd.err
python tools/rocpd_stats.py $OUT/kd/kd_results.db > $OUT/dense_kernel_stats.txt 2>&1
rm -rf $OUT/kd
grep -i "dense\|kernel " $OUT/dense_kernel_stats.txt | cut -c1-200
