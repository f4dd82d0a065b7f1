#!/bin/bash
# A/B/.. of N builds of the library on the SAME box: tools/exp/abn.sh ROUNDS A.so B.so ...   (files under stego_amd/lib/)
# prints step and forward-kernel us per run; leaves the LAST one installed as libstego_corr.so
L=stego_amd/lib
R=$1; shift
for i in $(seq $R); do
  for v in "$@"; do
    cp $L/$v $L/libstego_corr.so
    timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(1e3*d['ms_per_step'],2), d['roofline']['us_per_launch'], d['forward_backward_split'])"
  done
done
