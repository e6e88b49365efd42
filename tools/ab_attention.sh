#!/bin/bash
# same-box A/B of two builds of the library on the attention micro-benchmark (tools/bench_attention.py)
OTHER=${1:-clip_decontamination_amd/libsegearth_hip_old.so}
for i in 1 2 3; do
  SEGEARTH_HIP_LIB=$PWD/$OTHER ATTN_B=128 python tools/bench_attention.py 2>&1 | grep -v amdgpu | sed 's/^/other: /'
  ATTN_B=128 python tools/bench_attention.py 2>&1 | grep -v amdgpu | sed 's/^/tree:  /'
done
