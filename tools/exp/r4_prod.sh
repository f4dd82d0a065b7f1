#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_prod
mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "lazy_loss or on_device_rng or training_loop or loss_curve or total_api or cpp_autograd or without_the_torch_glue or captured_product" 2>&1 | tail -5 > $O/tests.txt
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/tests.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_prod/bench.json").read())
print("step us", round(1e3 * d["ms_per_step"], 2))
pp = d["product_path"]
for k in ("eager_ms_per_step", "eager_over_kernel_step", "graph_ms_per_step", "graph_over_kernel_step"):
    print(k, pp.get(k))
for k in ("fast_draws", "total_api", "total_api_fast_draws"):
    print(k, {kk: vv for kk, vv in pp[k].items() if kk != "what"})
PY
