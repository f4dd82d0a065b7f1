#!/bin/bash
# same-box A/B of library builds on the backbone forward: tools/exp/r4_vit_ab.sh ROUNDS name...   (stego_amd/lib/<name>.so)
export TMPDIR=/tmp
R=$1; shift
L=stego_amd/lib
cp $L/libstego_corr.so $L/_keep.so
for i in $(seq $R); do
  for v in "$@"; do
    cp $L/$v.so $L/libstego_corr.so
    for prec in f16x3 f16; do
      timeout 200 python tools/bench_vit.py --precision $prec --no-cpu --no-torch --iters 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$prec', 'ms', round(d['ms'],3))"
    done
  done
done
cp $L/_keep.so $L/libstego_corr.so
