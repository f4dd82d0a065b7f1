#!/bin/bash
# per-kernel times of the F16X3 backbone with parts removed (STEGO_DEBUG_VIT: 1 = no MFMAs, 2 = no stage copies after the first, 4 = no epilogue)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_vit; mkdir -p $O
for v in 0 4 1 2; do
  STEGO_DEBUG_VIT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks$v -o ks -- python tools/bench_vit.py --precision ${PREC:-f16x3} --no-cpu --no-torch --iters 3 > /dev/null 2>>$O/err.txt
  echo "== STEGO_DEBUG_VIT=$v" >> $O/ablate_${PREC:-f16x3}.txt
  python tools/rocpd_stats.py $(find $O/ks$v -name "*.db" | head -1) | grep "vit_gemm\|vit_attn\|layernorm" | cut -c1-110 >> $O/ablate_${PREC:-f16x3}.txt
  rm -rf $O/ks$v
done
cat $O/ablate_${PREC:-f16x3}.txt
