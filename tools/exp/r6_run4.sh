#!/bin/bash
# round 6: the early closing ticket (STEGO_DEBUG bit 4 = ticket at the end, as before): parity subset + same-process A/B at B = 32 / 16 + cfg-4
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06l}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $OUT/pytest_parity.txt 2>&1; tail -3 $OUT/pytest_parity.txt
for cfgb in "vits8_224 32" "vits8_224 16" "vitb8_320 32" "vitb8_320 16" "vits8_224 64"; do
  set -- $cfgb
  timeout 300 python tools/exp/r6_ab_debug.py $1 $2 0 4 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_ticket.txt
done
