#!/bin/bash
# round 6, visit: bench records at cfg-2 (B = 32), B = 16, cfg-4 B = 16 with the skeleton beside them
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06c}
mkdir -p $OUT
python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 200 --warmup 20 --batch 16 --no-cpu-baseline --no-alt > $OUT/bench_B16.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --workload vitb8_320 --batch 16 --no-cpu-baseline --no-alt > $OUT/bench_cfg4_vitb8_320_B16.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 20 --batch 8 --no-cpu-baseline --no-alt > $OUT/bench_B8.json 2>> $OUT/bench.err
tail -c 400 $OUT/bench.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1], "step us %.2f" % (1e3 * d["ms_per_step"]), "value %.0f" % d["value"], "fwd", {k: round(v, 2) for k, v in r["us_per_launch"].items()}, "frac %.3f" % r["frac"],
              "skeleton", r.get("skeleton", {}) and {k: r["skeleton"].get(k) for k in ("us", "other_layout_us", "error")}, "frac_of_skeleton", r.get("frac_of_skeleton"), d.get("forward_backward_split"), d.get("step_us_dist"))
    except Exception as e:
        print(f, "failed", e)
PY
