#!/bin/bash
# round 4: A/B of the lists-first backward's variants (STEGO_DEBUG_BWD bits): 2048 write-through DT stores, 4096 XCD-local unsample units
export TMPDIR=/tmp
O=gpurun_out/r4_bwd2
mkdir -p $O
for rep in 1 2; do
for v in 1024 0 2048 4096 6144; do
  STEGO_DEBUG_BWD=$v timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug_bwd=$v', 'step', round(1e3*d['ms_per_step'],2), 'fwd', round(d['roofline']['us_per_launch']['corr_fused_kernel'],2))" >> $O/ab.txt 2>&1
done
done
for v in 8 2056 4104 6152; do
  echo "--- STEGO_DEBUG_BWD=$v" >> $O/stamps.txt
  SDB=$v timeout 120 python tools/stamps_bwd_lists.py >> $O/stamps.txt 2>&1
done
STEGO_DEBUG_BWD=6144 timeout 600 python -m pytest tests/test_bwd_fused.py -x -q -m gpu 2>&1 | tail -3 > $O/test_6144.txt
cat $O/ab.txt $O/stamps.txt $O/test_6144.txt
