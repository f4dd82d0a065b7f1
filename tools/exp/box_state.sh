#!/bin/bash
# What kind of box is this?  Clocks / power state before, during and after a sustained fwd+bwd loop, next to the step times.
rocm-smi --showperflevel --showclocks --showpower --showmaxpower 2>/dev/null | grep -vE "^=|^$|WARNING" | head -24
( sleep 6; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | head -8 ) &
python bench.py --steps 60000 --warmup 20 --no-cpu-baseline --no-alt --launch eager 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('long run (60000 steps) us/step', round(1e3*d['ms_per_step'],2))"
wait
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('short run us/step', round(1e3*d['ms_per_step'],2), 'fused', round(d['roofline']['us_per_launch']['corr_fused_kernel'],1))"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt --fwd-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd-only us/step', round(1e3*d['ms_per_step'],2))"
