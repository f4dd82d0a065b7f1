#!/bin/bash
# round 6: the blockwise unmasked park (STEGO_DEBUG bit 32768 = every element masked, as before): parity + same-process A/B
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06o}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $OUT/pytest_parity.txt 2>&1; tail -2 $OUT/pytest_parity.txt
for cfgb in "vits8_224 32" "vits8_224 16" "vitb8_320 32" "vits8_224 64"; do
  set -- $cfgb
  timeout 300 python tools/exp/r6_ab_debug.py $1 $2 0 32768 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_park.txt
done
