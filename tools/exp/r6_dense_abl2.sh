#!/bin/bash
# compile-time ablations of dense_stream_kernel's whole-block chunk: build the harness per DS_ABL value HERE (no GPU needed), run them on the box
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DDS_ONLY6 -DDS_ABL=$a -I stego_amd/csrc -I include tools/ubench/dense_stream_bench.hip stego_amd/csrc/host_util.hip -o tools/ubench/bin/dense_stream_bench_abl$a &
done
wait
ls tools/ubench/bin/
