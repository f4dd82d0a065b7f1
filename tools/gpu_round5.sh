#!/bin/bash
# Round-5 evidence run on the GPU box: bench records (cfg-2, B = 16 / 64, cfg-4, F32 as other_precision), kernel stats of the bench command,
# PMC passes of the forward and the backward, stamps of both, KNN at 100 k, the box's state.   usage: tools/gpu_round5.sh <tag>   (writes gpurun_out/<tag>/...)
export TMPDIR=/tmp
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
(rocm-smi --showclocks --showperflevel; rocminfo | grep -i -E "compute unit|partition" | head -8) > $OUT/box.txt 2>&1
python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 200 --warmup 20 --batch 16 --no-cpu-baseline --no-alt > $OUT/bench_B16.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --batch 64 --no-cpu-baseline --no-alt > $OUT/bench_B64.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --workload vitb8_320 --no-cpu-baseline --no-alt > $OUT/bench_cfg4_vitb8_320.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --workload vitb8_320 --batch 16 --no-cpu-baseline --no-alt > $OUT/bench_cfg4_vitb8_320_B16.json 2>> $OUT/bench.err
python bench.py --steps 200 --warmup 20 --precision f32 --no-cpu-baseline --no-alt > $OUT/bench_f32.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt --launch eager > $OUT/ks_bench.json 2> $OUT/ks.err
python tools/rocpd_stats.py $OUT/ks/ks_results.db > $OUT/kernel_stats.txt 2>&1
bash tools/exp/pmc_fused.sh $OUT/pmc > /dev/null 2>&1
cp $OUT/pmc/summary.txt $OUT/pmc_summary.txt
python tools/stamps_fused.py > $OUT/stamps_fused.txt 2>&1
python tools/stamps_bwd_lists.py > $OUT/stamps_bwd_lists.txt 2>&1
python tools/bench_knn.py > $OUT/knn_100k.json 2>> $OUT/bench.err
find $OUT -name "*.db" -delete
rm -rf $OUT/ks; find $OUT/pmc -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d = json.loads(open(f).read())
        print(f.split("/")[-1], "step us %.2f" % (1e3 * d["ms_per_step"]), "value %.0f" % d["value"], "fwd", {k: round(v, 2) for k, v in d["roofline"]["us_per_launch"].items()}, "frac %.3f" % d["roofline"]["frac"], d.get("forward_backward_split"))
    except Exception as e:
        print(f, "failed", e)
PY
