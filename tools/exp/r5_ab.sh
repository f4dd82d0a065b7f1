#!/bin/bash
# round 5: same-box A/B of library variants + stamps + quick parity.  usage: r5_ab.sh TAG ROUNDS "variants to bench" "variants to stamp" "variants to parity-check"
export TMPDIR=/tmp
TAG=$1; R=$2; O=gpurun_out/$TAG
mkdir -p $O
L=stego_amd/lib
timeout 1200 bash tools/exp/abn.sh $R $3 > $O/ab.txt 2>&1
for v in $4; do
  cp $L/$v.so $L/libstego_corr.so
  echo "--- $v" >> $O/stamps.txt
  timeout 120 python tools/stamps_fused.py 2>&1 | grep -v amdgpu >> $O/stamps.txt
done
for v in $5; do
  cp $L/$v.so $L/libstego_corr.so
  echo "--- $v" >> $O/parity.txt
  timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden or full_size_cfg2 or stress_rotating or cfg4_vitb or give_up or fused_path_edge or batch_64" 2>&1 | tail -3 >> $O/parity.txt
done
cp $L/base.so $L/libstego_corr.so
cat $O/ab.txt; cat $O/parity.txt 2>/dev/null
