#!/bin/bash
# round 4: A/B of the backward's CT swizzle + parity + box state
export TMPDIR=/tmp
O=gpurun_out/r4_bwd3
mkdir -p $O
(rocm-smi --showclocks --showperflevel; rocminfo | grep -i -E "compute unit|marketing|wavefront|partition" | head -20) > $O/box.txt 2>&1
STAMPS=0 bash tools/exp/r4_ab.sh r4_bwd3 3 base swz > /dev/null 2>&1
cp stego_amd/lib/swz.so stego_amd/lib/libstego_corr.so
timeout 120 python tools/stamps_bwd_lists.py > $O/stamps_swz.txt 2>&1
timeout 900 python -m pytest tests/test_bwd_fused.py tests/test_parity_gpu.py -x -q -m gpu -k "bwd_fused or golden or full_size or edge or border or linear or above_72 or rounds_of_whole or training_loop or loss_curve or randomised or cpp_autograd or fp32_class" 2>&1 | tail -5 > $O/tests.txt
cat $O/ab.txt $O/stamps_swz.txt $O/tests.txt
