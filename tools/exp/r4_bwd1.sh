#!/bin/bash
# round 4: first GPU visit of the one-launch backward: parity, stamps, bench
export TMPDIR=/tmp
O=gpurun_out/r4_bwd1
mkdir -p $O
timeout 900 python -m pytest tests/test_bwd_fused.py -x -q -m gpu 2>&1 | tail -15 > $O/test_bwd_fused.txt
timeout 120 python tools/stamps_bwd_lists.py > $O/stamps_bwd_fused.txt 2>&1
timeout 120 python tools/stamps_bwd_lists.py 16 > $O/stamps_bwd_fused_B16.txt 2>&1
STEGO_DEBUG_BWD=1024 timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt > $O/bench_two_launch.json 2> $O/bench.err
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt > $O/bench.json 2>> $O/bench.err
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden or full_size or edge or border or linear or above_72 or rounds_of_whole or training_loop or loss_curve or randomised or cpp_autograd or stress or determin" 2>&1 | tail -8 > $O/test_parity_subset.txt
cat $O/test_bwd_fused.txt $O/stamps_bwd_fused.txt $O/stamps_bwd_fused_B16.txt $O/test_parity_subset.txt
python - <<'PY'
import json
for n in ("bench_two_launch", "bench"):
    try:
        d = json.loads(open("gpurun_out/r4_bwd1/%s.json" % n).read())
        print(n, "step us", round(1e3 * d["ms_per_step"], 2), d.get("forward_backward_split"))
    except Exception as e:
        print(n, "failed", e)
PY
