#!/bin/bash
# round 5, visit 1: store-pattern and gather-pipeline micro-benchmarks; A/B of the three-set gather pipeline (g3, g3h) against the base library
export TMPDIR=/tmp
O=gpurun_out/r5_1
mkdir -p $O
(rocm-smi --showclocks --showperflevel; rocminfo | grep -i -E "compute unit|partition" | head -8) > $O/box.txt 2>&1
timeout 120 tools/ubench/bin/wt_store > $O/wt_store.txt 2>&1
timeout 200 tools/ubench/bin/gather_lim > $O/gather_lim.txt 2>&1
timeout 900 bash tools/exp/abn.sh 2 base.so g3.so g3h.so > $O/ab.txt 2>&1
L=stego_amd/lib
cp $L/g3.so $L/libstego_corr.so
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/tests_g3.txt
timeout 120 python tools/stamps_fused.py 2>&1 | grep -v amdgpu > $O/stamps_g3.txt
cp $L/g3h.so $L/libstego_corr.so
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden or full_size_cfg2 or stress_rotating or cfg4_vitb" 2>&1 | tail -3 > $O/tests_g3h.txt
timeout 120 python tools/stamps_fused.py 2>&1 | grep -v amdgpu > $O/stamps_g3h.txt
cp $L/base.so $L/libstego_corr.so
cat $O/wt_store.txt; grep heavy $O/gather_lim.txt; grep pipe3 $O/gather_lim.txt; cat $O/ab.txt $O/tests_g3.txt $O/tests_g3h.txt
