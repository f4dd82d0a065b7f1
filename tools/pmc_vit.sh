#!/bin/bash
# PMC passes over the backbone forward (separate runs, kernel-trace only): HBM bytes, L2 hit rate, MFMA busy.
export TMPDIR=/tmp
OUT=gpurun_out/vitpmc
mkdir -p $OUT
for CNT in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  NAME=$(echo $CNT | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $CNT -d $OUT/pmc_$NAME -o pmc -- python tools/bench_vit.py --no-cpu --no-torch --iters 2 > /dev/null 2> $OUT/pmc_$NAME.err
  echo "== $CNT"
  python tools/rocpd_stats.py $OUT/pmc_$NAME/pmc_results.db 2>&1 | grep -E "counter|vit_" | grep -v "^3vit.*GemmParams.* [0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+" 
done
find $OUT -name "*.db" -delete
