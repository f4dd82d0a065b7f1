#!/bin/bash
# round 4: A/B by STEGO_DEBUG_BWD values ($@) of the installed library, then stamps and the backward tests at the last value
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r4_bwd4}
mkdir -p $O
for rep in 1 2 3; do
for v in "$@"; do
  STEGO_DEBUG_BWD=$v timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt ${B:+--batch $B} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug_bwd=$v', 'step', round(1e3*d['ms_per_step'],2), 'fwd', round(d['roofline']['us_per_launch']['corr_fused_kernel'],2))" >> $O/ab.txt 2>&1
done
done
for v in "$@"; do
  echo "--- STEGO_DEBUG_BWD=$v + 8" >> $O/stamps.txt
  SDB=$((v + 8)) timeout 120 python tools/stamps_bwd_lists.py 2>&1 | grep -v amdgpu >> $O/stamps.txt
done
timeout 900 python -m pytest tests/test_bwd_fused.py tests/test_parity_gpu.py -x -q -m gpu -k "bwd_fused or golden or full_size or edge or border or linear or above_72 or training_loop or loss_curve or randomised or fp32_class" 2>&1 | tail -3 > $O/tests.txt
cat $O/ab.txt $O/stamps.txt $O/tests.txt
