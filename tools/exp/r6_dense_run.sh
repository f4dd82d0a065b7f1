#!/bin/bash
# round 6: the streaming dense-correspondence kernel: parity tests, timing (stream vs STEGO_DEBUG bit 21 = row-block kernel), kernel stats
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06g}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_dense_corr.py -x -q -m gpu > $OUT/pytest_dense.txt 2>&1; tail -5 $OUT/pytest_dense.txt
timeout 300 python tools/bench_dense.py > $OUT/dense.json 2> $OUT/dense.err; cut -c1-330 $OUT/dense.json
STEGO_DEBUG=2097152 timeout 300 python tools/bench_dense.py > $OUT/dense_rowblock.json 2>> $OUT/dense.err; cut -c1-330 $OUT/dense_rowblock.json
rocprofv3 --kernel-trace --stats -d $OUT/kd -o kd -- python tools/bench_dense.py > /dev/null 2> $OUT/kd.err
python tools/rocpd_stats.py $OUT/kd/kd_results.db > $OUT/dense_kernel_stats.txt 2>&1
rm -rf $OUT/kd
grep -i "dense\|kernel " $OUT/dense_kernel_stats.txt | cut -c1-200
