#!/bin/bash
# One GPU-box visit: smoke, parity tests, ablation matrix, bench, rocprof kernel stats + PMC passes.
# Usage: tools/gpu_round.sh <tag> [stages...]   stages: smoke test ablate bench prof pmc (default: all)
TAG=${1:-r01}
shift
STAGES=${@:-smoke test ablate bench prof pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
: > $OUT/summary.txt
if has smoke; then
  echo "== smoke" | tee -a $OUT/summary.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
  tail -3 $OUT/smoke.log | tee -a $OUT/summary.txt
fi
if has test; then
  echo "== pytest -m gpu" | tee -a $OUT/summary.txt
  timeout 1500 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -40 | tee -a $OUT/summary.txt
  grep -E "AssertionError" $OUT/pytest.log | head -30 | tee -a $OUT/summary.txt
fi
if has ablate; then
  echo "== ablation matrix" | tee -a $OUT/summary.txt
  timeout 600 python tools/ablate.py > $OUT/ablate.log 2>&1; echo "ablate rc=$?" | tee -a $OUT/summary.txt
  grep -E "^\{'" $OUT/ablate.log | tee -a $OUT/summary.txt
  tail -3 $OUT/ablate.log | grep -v "^{" | tee -a $OUT/summary.txt
fi
if has bench; then
  echo "== bench" | tee -a $OUT/summary.txt
  timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
  cat $OUT/bench.json | tee -a $OUT/summary.txt; tail -3 $OUT/bench.err | tee -a $OUT/summary.txt
  timeout 300 python bench.py --steps 200 --warmup 20 --precision f32 --no-cpu-baseline --no-alt > $OUT/bench_f32.json 2>> $OUT/bench.err
  cat $OUT/bench_f32.json | tee -a $OUT/summary.txt
  timeout 300 python bench.py --steps 200 --warmup 20 --fwd-only --no-cpu-baseline --no-alt > $OUT/bench_fwd.json 2>> $OUT/bench.err
  cat $OUT/bench_fwd.json | tee -a $OUT/summary.txt
fi
if has prof; then
  echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-graph --no-alt > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?" | tee -a $OUT/summary.txt
  python tools/rocpd_stats.py $OUT/prof/bench_results.db 2>&1 | head -12 | tee -a $OUT/summary.txt
fi
if has pmc; then
  echo "== rocprofv3 PMC passes (separate runs)" | tee -a $OUT/summary.txt
  rocprofv3 -L > $OUT/counters_list.txt 2>&1
  for CNT in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    NAME=$(echo $CNT | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --kernel-trace --pmc $CNT -d $OUT/pmc_$NAME -o pmc -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-graph --no-alt > /dev/null 2> $OUT/pmc_$NAME.err; echo "pmc $CNT rc=$?" | tee -a $OUT/summary.txt
    python tools/rocpd_stats.py $OUT/pmc_$NAME/pmc_results.db 2>&1 | grep -E "counter|corr_|sample_norm|knn_" | tee -a $OUT/summary.txt
    rm -f $OUT/pmc_$NAME/pmc_results.db.keep
  done
fi
# keep the merged payload small: the raw sqlite databases are summarised above
find $OUT -name "*.db" -size +20M -delete
