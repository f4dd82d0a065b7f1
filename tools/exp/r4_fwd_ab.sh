#!/bin/bash
# round 4: A/B by STEGO_DEBUG (forward) values ($@) of the installed library; then stamps at each value (+256) and forward parity
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r4_fwd_ab}
mkdir -p $O
for rep in 1 2 3; do
for v in "$@"; do
  STEGO_DEBUG=$v timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt ${B:+--batch $B} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug=$v', 'step', round(1e3*d['ms_per_step'],2), 'fwd', round(d['roofline']['us_per_launch']['corr_fused_kernel'],2))" >> $O/ab.txt 2>&1
done
done
for v in "$@"; do
  echo "--- STEGO_DEBUG=$v + 256" >> $O/stamps.txt
  timeout 120 python tools/stamps_fused.py $((v + 256)) 2>&1 | grep -v amdgpu >> $O/stamps.txt
done
if [ -n "$PARITY" ]; then
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "full_size_cfg2 or cfg4_vitb or stress_rotating or give_up or shared_device_mode or fused_path_edge or rounds_of_whole or code_dimensions_above_72_forward or batch_64 or foreign_kernel or largest_batch or golden" 2>&1 | tail -3 > $O/parity.txt
fi
cat $O/ab.txt $O/stamps.txt $O/parity.txt 2>/dev/null
