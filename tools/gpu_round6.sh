#!/bin/bash
# Round-6 evidence run on the GPU box: bench records (cfg-2 with the in-run counter passes and the skeleton; B = 16 / 8 / 64; cfg-4 B = 32 / 16;
# F32), kernel stats of the bench command at B = 32 and B = 16, PMC passes of forward + backward, the traffic skeleton at four shapes, the
# MFMA-rate micro-benchmark, stamps of the full-tile and the column-half kernels, half vs full launch of one library, the dense correspondence
# (bench + stamps of dense_stream_kernel), KNN at 100 k, same-process A/Bs of the early closing ticket (bit 4) and the twelve-wave half kernel (bit 2).
# usage: tools/gpu_round6.sh <tag>   (writes gpurun_out/<tag>/...)
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
(rocm-smi --showclocks --showperflevel; rocminfo | grep -i -E "compute unit|partition" | head -8) > $OUT/box.txt 2>&1
python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 200 --warmup 20 --batch 16 --no-cpu-baseline --no-alt > $OUT/bench_B16.json 2>> $OUT/bench.err
python bench.py --steps 200 --warmup 20 --batch 8 --no-cpu-baseline --no-alt --traffic none > $OUT/bench_B8.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --batch 64 --no-cpu-baseline --no-alt --traffic none > $OUT/bench_B64.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --workload vitb8_320 --no-cpu-baseline --no-alt --traffic none > $OUT/bench_cfg4_vitb8_320.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --workload vitb8_320 --batch 16 --no-cpu-baseline --no-alt > $OUT/bench_cfg4_vitb8_320_B16.json 2>> $OUT/bench.err
python bench.py --steps 200 --warmup 20 --precision f32 --no-cpu-baseline --no-alt --traffic none > $OUT/bench_f32.json 2>> $OUT/bench.err
python bench.py --steps 200 --warmup 20 --batch 16 --precision f32 --no-cpu-baseline --no-alt --traffic none > $OUT/bench_B16_f32.json 2>> $OUT/bench.err
for B in 32 16; do
  rocprofv3 --kernel-trace --stats -d $OUT/ks$B -o ks -- python bench.py --steps 100 --warmup 10 --batch $B --no-cpu-baseline --no-alt --launch eager --traffic none > $OUT/ks_bench_B$B.json 2> $OUT/ks$B.err
  python tools/rocpd_stats.py $OUT/ks$B/ks_results.db > $OUT/kernel_stats_B$B.txt 2>&1
done
bash tools/exp/pmc_fused.sh $OUT/pmc > /dev/null 2>&1
cp $OUT/pmc/summary.txt $OUT/pmc_summary.txt
B=16 bash tools/exp/pmc_fused.sh $OUT/pmc16 > /dev/null 2>&1
cp $OUT/pmc16/summary.txt $OUT/pmc_summary_B16.txt
python tools/stamps_fused.py > $OUT/stamps_fused.txt 2>&1
python tools/stamps_half.py > $OUT/stamps_half_B16.txt 2>&1
DBG=1 python tools/stamps_half.py > $OUT/stamps_half_B16_no_mfma.txt 2>&1
WL=vitb8_320 python tools/stamps_half.py > $OUT/stamps_half_cfg4_B16.txt 2>&1
python tools/stamps_bwd_lists.py > $OUT/stamps_bwd_lists.txt 2>&1
(python tools/exp/r6_half_check.py vits8_224 16 8 4; python tools/exp/r6_half_check.py vitb8_320 16 8) 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $OUT/half_vs_full.txt
for shp in "32 384 28" "16 384 28" "32 768 40" "16 768 40"; do
  set -- $shp
  stego_amd/lib/fused_skeleton.bin $1 $2 $3 40 > $OUT/skeleton_B$1_C$2.txt 2>&1
done
SKEL_MODES=1 stego_amd/lib/fused_skeleton.bin 32 384 28 30 > $OUT/skeleton_B32_hot_cold.txt 2>&1
tools/ubench/bin/mfma_rate > $OUT/ubench_mfma_rate.txt 2>&1
python tools/bench_dense.py > $OUT/dense_corr.json 2>> $OUT/bench.err
python tools/bench_knn.py > $OUT/knn_100k.json 2>> $OUT/bench.err
python tools/exp/r6_dense_stamps.py > $OUT/stamps_dense_stream.txt 2>&1
(python tools/exp/r6_ab_debug.py vits8_224 32 0 4; python tools/exp/r6_ab_debug.py vits8_224 16 0 4 2) 2>&1 | grep -v amdgpu.ids > $OUT/ab_ticket_and_waves.txt
find $OUT -name "*.db" -delete
rm -rf $OUT/ks32 $OUT/ks16; find $OUT/pmc $OUT/pmc16 -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1], "step us %.2f" % (1e3 * d["ms_per_step"]), "value %.0f" % d["value"], "fwd", {k: round(v, 2) for k, v in r["us_per_launch"].items()}, "frac %.3f" % r["frac"],
              "skeleton", (r.get("skeleton") or {}).get("us"), "frac_of_skeleton", r.get("frac_of_skeleton"), "traffic", r.get("traffic"), (r.get("traffic_source") or "")[:40], d.get("forward_backward_split"))
    except Exception as e:
        print(f, "failed", e)
PY
