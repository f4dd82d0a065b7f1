#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-cached}; mkdir -p $OUT
python tools/exp/cached_step_loop.py 100 > $OUT/enqueue.txt 2>&1
cd /tmp; rm -rf /tmp/cs
rocprofv3 --kernel-trace --stats -d /tmp/cs -o cs -- python $GRAFT_REPO_ROOT/tools/exp/cached_step_loop.py 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py /tmp/cs/cs_results.db > $OUT/kernel_stats.txt 2>&1
