#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3v7; mkdir -p $OUT
bash tools/exp/abn.sh 3 base.so rounds.so 2>&1 | tee $OUT/ab.txt
cp stego_amd/lib/rounds.so stego_amd/lib/libstego_corr.so
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "rounds or batch_64 or largest or stress or golden" 2>&1 | tail -3
