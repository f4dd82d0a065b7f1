#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; V=$2; shift; shift
cp stego_amd/lib/$V.so stego_amd/lib/libstego_corr.so
timeout 200 python tools/stamps_fused.py "$@" 2>&1 | grep -v amdgpu > $O/stamps.txt
cp stego_amd/lib/base.so stego_amd/lib/libstego_corr.so
cat $O/stamps.txt | grep -E "debug=|anchor ready|gather head|main loop end|   end |ring loop \(|light slots|intra slots|gathered slots  |last workgroup"
