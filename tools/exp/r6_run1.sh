#!/bin/bash
# round 6, visit 1: the traffic skeleton (FULL vs HALF layouts) at cfg-2, B = 16 and cfg-4; the bench line with the new fields
export TMPDIR=/tmp
OUT=gpurun_out/r06a
mkdir -p $OUT
(rocm-smi --showclocks --showperflevel; rocminfo | grep -i -E "compute unit|partition" | head -8) > $OUT/box.txt 2>&1
tools/ubench/bin/fused_skeleton 32 384 28 40 > $OUT/skeleton_B32.txt 2>&1
tools/ubench/bin/fused_skeleton 16 384 28 40 > $OUT/skeleton_B16.txt 2>&1
tools/ubench/bin/fused_skeleton 32 768 40 30 > $OUT/skeleton_cfg4.txt 2>&1
python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
cat $OUT/skeleton_B32.txt
