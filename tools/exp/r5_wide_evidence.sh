#!/bin/bash
# Evidence of the feature_samples 12 .. 16 path (csrc/corr_wide.hip): bench records, kernel stats of the product step, ablations, per-step host / total times.
#   usage (GPU box): bash tools/exp/r5_wide_evidence.sh <tag>      -> gpurun_out/<tag>/...
TAG=${1:-r05g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 100 --warmup 20 --feature-samples 16 --no-cpu-baseline --no-alt > $OUT/bench_S16.json 2> $OUT/err.txt
python bench.py --steps 100 --warmup 20 --feature-samples 12 --no-cpu-baseline --no-alt > $OUT/bench_S12.json 2>> $OUT/err.txt
python bench.py --steps 100 --warmup 20 --feature-samples 16 --batch 16 --no-cpu-baseline --no-alt > $OUT/bench_S16_B16.json 2>> $OUT/err.txt
for S in 16 12; do
  bash tools/exp/r5_generic_prof.sh $S > /dev/null 2>&1
  cp gpurun_out/r5_gen_$S/kernel_stats.txt $OUT/kernel_stats_S$S.txt
done
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/r5_gen_16/ks gpurun_out/r5_gen_12/ks
python tools/exp/r5_rowblock_abl.py > $OUT/rowblock_abl.txt 2>&1
python tools/exp/r5_wide_bwd_abl.py 16 > $OUT/bwd_abl.txt 2>&1
python tools/exp/generic_steps.py 16 12 > $OUT/steps.txt 2>&1
python tools/exp/generic_time.py > $OUT/product_api_time.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d = json.loads(open(f).read())
        print(f.split("/")[-1], "step us %.1f" % (1e3 * d["ms_per_step"]), "value %.0f" % d["value"], d["roofline"]["us_per_launch"], "frac %.3f" % d["roofline"]["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
