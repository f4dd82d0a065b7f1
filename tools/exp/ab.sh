#!/bin/bash
# A/B of two builds of the library on the SAME box (box-to-box variance is larger than most kernel changes):
# usage: tools/exp/ab.sh A.so B.so [rounds]   (files under stego_amd/lib/); prints step and forward-kernel us per run
L=stego_amd/lib
R=${3:-2}
for i in $(seq $R); do
  for v in $1 $2; do
    cp $L/$v $L/libstego_corr.so
    timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(1e3*d['ms_per_step'],2), d['roofline']['us_per_launch'])"
  done
done
cp $L/$2 $L/libstego_corr.so
