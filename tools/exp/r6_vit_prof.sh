#!/bin/bash
# round 6: per-kernel times of the frozen backbone (tools/bench_vit.py --precision f16x3, B = 64 ViT-S/8) + the debug-bit ablations of its GEMM (4 = no epilogue, 1 = no MFMAs)
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06n}
mkdir -p $OUT
for d in 0 4 1; do
  STEGO_DEBUG_VIT=$d rocprofv3 --kernel-trace --stats -d $OUT/kv$d -o kv -- python tools/bench_vit.py --precision f16x3 --no-cpu --no-torch > $OUT/vit_$d.json 2> $OUT/kv.err
  echo "== STEGO_DEBUG_VIT=$d $(cut -c1-160 $OUT/vit_$d.json)" | tee -a $OUT/vit_kernels.txt
  python tools/rocpd_stats.py $OUT/kv$d/kv_results.db | grep -i "vit" | cut -c1-110 | tee -a $OUT/vit_kernels.txt
  rm -rf $OUT/kv$d
done
