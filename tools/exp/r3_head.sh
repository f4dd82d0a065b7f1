#!/bin/bash
# head kernels: parity tests, per-kernel stats of forward + backward at 2B = 64, the cached-token step
export TMPDIR=/tmp
OUT=gpurun_out/${1:-head}; mkdir -p $OUT
python -m pytest tests/test_head_native.py -x -q -m gpu 2>&1 | tail -5 > $OUT/tests.txt
cd /tmp; rm -rf /tmp/hk
rocprofv3 --kernel-trace --stats -d /tmp/hk -o hk -- python $GRAFT_REPO_ROOT/tools/exp/head_loop.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py /tmp/hk/hk_results.db > $OUT/head_kernel_stats.txt 2>&1
python tools/bench_step.py > $OUT/bench_step.json 2> $OUT/bench_step.err
