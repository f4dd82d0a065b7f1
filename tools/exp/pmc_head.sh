#!/bin/bash
# PMC passes over the native head's kernels (one counter group per pass, kernel-trace only).  usage: tools/exp/pmc_head.sh <outdir>
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_head}
mkdir -p $OUT
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  NAME=$(echo $CNT | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $CNT -d $OUT/pmc_$NAME -o pmc -- python tools/exp/head_loop.py 3 > /dev/null 2> $OUT/pmc_$NAME.err
  python tools/rocpd_stats.py $OUT/pmc_$NAME/pmc_results.db 2>&1 | grep -E "counter|head_gemm|head_wgrad" | cut -c1-260 | tee -a $OUT/summary.txt
done
find $OUT -name "*.db" -delete
