export TMPDIR=/tmp
OUT=gpurun_out/r01f
mkdir -p $OUT
for CNT in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  NAME=$(echo $CNT | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $CNT -d $OUT/pmc_$NAME -o pmc -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-graph --fwd-only > /dev/null 2> $OUT/pmc_$NAME.err
  python tools/rocpd_stats.py $OUT/pmc_$NAME/pmc_results.db 2>&1 | grep -E "counter|sample_norm|corr_tile" 
done
find $OUT -name "*.db" -delete
