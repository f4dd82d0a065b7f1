#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r10
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "draws or captured_product or cpp_autograd" 2>&1 | tail -15 > gpurun_out/r10/tests.txt
python tools/exp/profile_eager.py > gpurun_out/r10/profile_eager.txt 2>&1
python tools/host_overhead.py > gpurun_out/r10/host_overhead.txt 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r10/bench.json 2> gpurun_out/r10/bench.err
python bench.py --steps 800 --warmup 20 --no-cpu-baseline > gpurun_out/r10/bench800.json 2>> gpurun_out/r10/bench.err
