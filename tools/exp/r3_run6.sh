#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3v6; mkdir -p $OUT
cp stego_amd/lib/rounds.so stego_amd/lib/libstego_corr.so
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q 2>&1 | tail -5 | tee $OUT/pytest.txt
bash tools/exp/abn.sh 2 base.so rounds.so 2>&1 | tee $OUT/ab.txt
for v in base rounds; do
cp stego_amd/lib/$v.so stego_amd/lib/libstego_corr.so
timeout 200 python bench.py --batch 64 --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v B=64', round(1e3*d['ms_per_step'],2), d['roofline'].get('us_per_launch'), round(d['roofline']['frac'],3), d['value'])" | tee -a $OUT/ab.txt
done
cp stego_amd/lib/rounds.so stego_amd/lib/libstego_corr.so
