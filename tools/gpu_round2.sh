#!/bin/bash
# Round-2 evidence run on the GPU box: kernel stats of the bench command, PMC passes of the fused forward, stamps.
# usage: tools/gpu_round2.sh <tag>   (writes gpurun_out/<tag>/...)
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt --launch eager > $OUT/ks_bench.json 2> $OUT/ks.err
python tools/rocpd_stats.py $OUT/ks/ks_results.db > $OUT/kernel_stats.txt 2>&1
bash tools/exp/pmc_fused.sh $OUT/pmc > /dev/null 2>&1
cp $OUT/pmc/summary.txt $OUT/pmc_summary.txt
python tools/stamps_fused.py > $OUT/stamps_fused.txt 2>&1
find $OUT -name "*.db" -delete
rm -rf $OUT/ks $OUT/pmc/pmc_*
