#!/bin/bash
# same-box A/B of library variants at a batch size: r5_abB.sh B ROUNDS v1.so v2.so ...
export TMPDIR=/tmp
L=stego_amd/lib; B=$1; R=$2; shift; shift
for i in $(seq $R); do
  for v in "$@"; do
    cp $L/$v $L/libstego_corr.so
    timeout 200 python bench.py --steps 200 --warmup 20 --batch $B --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B $v', round(1e3*d['ms_per_step'],2), d['roofline']['us_per_launch'])"
  done
done
cp $L/base.so $L/libstego_corr.so
