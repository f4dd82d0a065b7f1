#!/bin/bash
# round 4, run 1: what bounds the ring loop's gather (tools/ubench/gather_lim), timeline + stamps + bench of the round-3 kernel
export TMPDIR=/tmp
O=gpurun_out/r4_1
mkdir -p $O
timeout 120 tools/ubench/bin/gather_lim > $O/gather_lim.txt 2>&1
L=stego_amd/lib
cp $L/tl.so $L/libstego_corr.so
timeout 120 python tools/timeline_fused.py > $O/timeline.txt 2>&1
cp $L/base.so $L/libstego_corr.so
timeout 120 python tools/stamps_fused.py > $O/stamps.txt 2>&1
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --steps 200 --warmup 20 --batch 16 --no-cpu-baseline --no-alt > $O/bench_B16.json 2>> $O/bench.err
rocm-smi --showclocks --showperflevel > $O/smi.txt 2>&1
