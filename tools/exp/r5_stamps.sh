#!/bin/bash
# round 5: stamps of a list of library variants (ring loop percentiles etc.) + forward-kernel time.  usage: r5_stamps.sh TAG v1 v2 ...
export TMPDIR=/tmp
TAG=$1; shift; O=gpurun_out/$TAG
mkdir -p $O
L=stego_amd/lib
for v in "$@"; do
  cp $L/$v.so $L/libstego_corr.so
  echo "--- $v" >> $O/stamps.txt
  timeout 120 python tools/stamps_fused.py 2>&1 | grep -v amdgpu >> $O/stamps.txt
  timeout 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-alt --fwd-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(1e3*d['ms_per_step'],2), d['roofline']['us_per_launch'])" >> $O/fwd.txt
done
cp $L/base.so $L/libstego_corr.so
grep -E "^---|ring loop \(|main loop end|anchor ready|   end  " $O/stamps.txt; cat $O/fwd.txt
