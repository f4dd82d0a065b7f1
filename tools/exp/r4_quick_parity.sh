#!/bin/bash
# quick forward parity of a library variant: tools/exp/r4_quick_parity.sh OUT name   (installs stego_amd/lib/<name>.so for the run)
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
L=stego_amd/lib
cp $L/libstego_corr.so $L/_keep2.so
cp $L/$2.so $L/libstego_corr.so
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "full_size_cfg2 or cfg4_vitb or stress_rotating or give_up or shared_device_mode or fused_path_edge or rounds_of_whole or code_dimensions_above_72_forward or batch_64" 2>&1 | tail -8 > $O/parity_$2.txt
cp $L/_keep2.so $L/libstego_corr.so
cat $O/parity_$2.txt
