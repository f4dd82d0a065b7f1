#!/bin/bash
# same-box A/B/.. of library builds (stego_amd/lib/<name>.so): tools/exp/r4_ab.sh OUT ROUNDS name...   -> gpurun_out/OUT/{ab.txt,stamps_<name>.txt}
# STAMPS=0 skips the stamps; B=16 benches another batch
export TMPDIR=/tmp
O=gpurun_out/$1; R=$2; shift; shift
mkdir -p $O
L=stego_amd/lib
cp $L/libstego_corr.so $L/_keep.so
for i in $(seq $R); do
  for v in "$@"; do
    cp $L/$v.so $L/libstego_corr.so
    timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt ${B:+--batch $B} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'step', round(1e3*d['ms_per_step'],2), 'fwd', round(d['roofline']['us_per_launch']['corr_fused_kernel'],2))" >> $O/ab.txt 2>&1
  done
done
if [ "${STAMPS:-1}" != "0" ]; then
  for v in "$@"; do
    cp $L/$v.so $L/libstego_corr.so
    timeout 120 python tools/stamps_fused.py > $O/stamps_$v.txt 2>&1
  done
fi
cp $L/_keep.so $L/libstego_corr.so
cat $O/ab.txt
