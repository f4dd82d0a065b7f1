#!/bin/bash
# rocprofv3 kernel stats of the generic (feature_samples > 11) loss step: S = $1 (default 16)
S=${1:-16}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_gen_$S
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python $GRAFT_REPO_ROOT/tools/exp/generic_loop.py $S 20 > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $OUT/ks -name "*results.db" | head -1) > $OUT/kernel_stats.txt 2>&1 || ls -R $OUT | head -30
head -40 $OUT/kernel_stats.txt
