#!/bin/bash
# round 5: diagnostics of one variant: operand-image diff against the fallback statement, stamps (given debug values), quick parity, bench A/B vs others
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
V=$2
cp stego_amd/lib/$V.so stego_amd/lib/libstego_corr.so
timeout 200 python tools/exp/p1_diff.py 2>&1 | grep -v amdgpu | cut -c1-400 > $O/p1_diff.txt
timeout 200 python tools/stamps_fused.py ${STAMPS:-1280} 2>&1 | grep -v amdgpu > $O/stamps.txt
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden or full_size_cfg2 or stress_rotating or cfg4_vitb or give_up or fused_path_edge or batch_64 or foreign_kernel or shared_device" 2>&1 | tail -15 > $O/parity.txt
timeout 900 bash tools/exp/abn.sh 2 $3 > $O/ab.txt 2>&1
cp stego_amd/lib/base.so stego_amd/lib/libstego_corr.so
grep -E "^seed|differing" $O/p1_diff.txt; cat $O/stamps.txt | grep -E "anchor ready|main loop end|   end |ring loop \(|light slots|intra slots|gathered slots  "; cat $O/parity.txt | tail -6; cat $O/ab.txt
