#!/bin/bash
# GPU visit: captured one-launch draws (tests), host profile of the eager product path, bench record
export TMPDIR=/tmp
mkdir -p gpurun_out/r8
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "draws or captured_product" 2>&1 | tail -15 > gpurun_out/r8/tests.txt
python tools/exp/profile_eager.py > gpurun_out/r8/profile_eager.txt 2>&1
python tools/host_overhead.py > gpurun_out/r8/host_overhead.txt 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r8/bench.json 2> gpurun_out/r8/bench.err
