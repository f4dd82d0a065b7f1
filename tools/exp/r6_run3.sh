#!/bin/bash
# round 6, visit: the sixteen-wave form of the column-half kernel (C = 384): parity subset + same-process A/B against the twelve-wave form (STEGO_DEBUG bit 2)
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06d}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "column_half" > $OUT/pytest_half.txt 2>&1; tail -5 $OUT/pytest_half.txt
timeout 300 python tools/exp/r6_ab_debug.py vits8_224 16 0 2 > $OUT/ab_B16.txt 2>&1; cat $OUT/ab_B16.txt
timeout 300 python tools/exp/r6_ab_debug.py vits8_224 8 0 2 > $OUT/ab_B8.txt 2>&1; cat $OUT/ab_B8.txt
