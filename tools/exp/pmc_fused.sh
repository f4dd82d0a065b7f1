#!/bin/bash
# PMC passes over the fused forward and the lists-first backward (one counter group per pass, no other tracing).  usage: tools/exp/pmc_fused.sh <outdir>
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_fused}
mkdir -p $OUT
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  NAME=$(echo $CNT | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $CNT -d $OUT/pmc_$NAME -o pmc -- env BWD=1 python tools/exp/fwd_loop.py 12 > /dev/null 2> $OUT/pmc_$NAME.err
  python tools/rocpd_stats.py $OUT/pmc_$NAME/pmc_results.db 2>&1 | grep -E "counter|corr_fused|corr_bwd_tile_build|corr_unsample_list" | tee -a $OUT/summary.txt
done
find $OUT -name "*.db" -delete
