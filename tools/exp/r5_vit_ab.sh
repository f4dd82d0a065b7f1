#!/bin/bash
# same-box A/B of two library builds on the backbone bench: usage r5_vit_ab.sh A.so B.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in $A $B; do
    STEGO_LIB_PATH=$GRAFT_REPO_ROOT/stego_amd/lib/$L python tools/bench_vit.py --precision f16x3 --no-cpu --no-torch --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'f16x3', round(d['ms'],3))"
    STEGO_LIB_PATH=$GRAFT_REPO_ROOT/stego_amd/lib/$L python tools/bench_vit.py --precision f16 --no-cpu --no-torch --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'f16  ', round(d['ms'],3))"
  done
done
