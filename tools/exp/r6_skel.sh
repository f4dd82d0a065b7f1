#!/bin/bash
# round 6: the traffic skeleton at the shapes given as "B C HW" triples (default: cfg-2)   usage: tools/exp/r6_skel.sh <tag> ["32 384 28" ...]
export TMPDIR=/tmp
OUT=gpurun_out/$1
shift
mkdir -p $OUT
if [ $# -eq 0 ]; then set -- "32 384 28"; fi
for shp in "$@"; do
    set -- $shp
    tools/ubench/bin/fused_skeleton $1 $2 $3 40 > $OUT/skeleton_B$1_C$2.txt 2>&1
    cat $OUT/skeleton_B$1_C$2.txt
done
