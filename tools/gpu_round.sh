#!/bin/bash
# One GPU-box visit: smoke, parity tests, bench, rocprof kernel stats.  Usage: tools/gpu_round.sh <tag> [pytest-args]
TAG=${1:-r01}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/smoke.log | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -40 $OUT/pytest.log | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt; tail -5 $OUT/bench.err | tee -a $OUT/summary.txt
timeout 300 python bench.py --steps 100 --warmup 10 --fwd-only --no-cpu-baseline > $OUT/bench_fwd.json 2>> $OUT/bench.err
cat $OUT/bench_fwd.json | tee -a $OUT/summary.txt
echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-graph > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats*" | head -3 | while read f; do echo $f; head -12 "$f"; done | tee -a $OUT/summary.txt
