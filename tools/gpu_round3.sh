#!/bin/bash
# Round-3 evidence run on the GPU box: bench records (cfg-2, B = 16 / 64, cfg-4, F32), kernel stats of the bench command, PMC passes
# of the fused forward, its stamps, KNN at 100 k.   usage: tools/gpu_round3.sh <tag>   (writes gpurun_out/<tag>/...)
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 200 --warmup 20 --batch 16 --no-cpu-baseline --no-alt > $OUT/bench_B16.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --batch 64 --no-cpu-baseline --no-alt > $OUT/bench_B64.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --workload vitb8_320 --no-cpu-baseline > $OUT/bench_cfg4_vitb8_320.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --workload vitb8_320 --batch 16 --no-cpu-baseline --no-alt > $OUT/bench_cfg4_vitb8_320_B16.json 2>> $OUT/bench.err
python tools/bench_knn.py > $OUT/knn_100k.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt --launch eager > $OUT/ks_bench.json 2> $OUT/ks.err
python tools/rocpd_stats.py $OUT/ks/ks_results.db > $OUT/kernel_stats.txt 2>&1
bash tools/exp/pmc_fused.sh $OUT/pmc > /dev/null 2>&1
cp $OUT/pmc/summary.txt $OUT/pmc_summary.txt
python tools/stamps_fused.py > $OUT/stamps_fused.txt 2>&1
python tools/stamps_bwd.py > $OUT/stamps_bwd.txt 2>&1
find $OUT -name "*.db" -delete
rm -rf $OUT/ks $OUT/pmc/pmc_*
