#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_bwd5
mkdir -p $O
timeout 900 python -m pytest tests/test_bwd_fused.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
for rep in 1 2; do
for v in 1024 0; do
  STEGO_DEBUG_BWD=$v timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt --precision f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 debug_bwd=$v', 'step', round(1e3*d['ms_per_step'],2), 'fwd', round(d['roofline']['us_per_launch']['corr_fused_kernel'],2))" >> $O/ab.txt 2>&1
done
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt --batch 16 > $O/bench_B16.json 2>/dev/null
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-alt --workload vitb8_320 > $O/bench_cfg4.json 2>/dev/null
cat $O/tests.txt $O/ab.txt
python - <<'PY'
import json
for n in ("bench_B16", "bench_cfg4"):
    d = json.loads(open("gpurun_out/r4_bwd5/%s.json" % n).read())
    print(n, "step us", round(1e3 * d["ms_per_step"], 2), "fwd", d["roofline"]["us_per_launch"], "frac", round(d["roofline"]["frac"], 3))
PY
