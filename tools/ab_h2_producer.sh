#!/bin/bash
# What does the two-plane folded-LayerNorm producer epilogue (gemm_h2_persist<1, 2>) pay for its two-plane copy and its slice statistics?
# Builds three ablated copies of the library (-DSG_H2_PROD_ABL=n in gemm_bf16.hip: no copy / no statistics / neither -- WRONG results by
# design) and runs bench.py --precision f16x2 with each on the same box.
#   tools/ab_h2_producer.sh build            (in the build container)
#   tools/ab_h2_producer.sh run              (on the GPU box)
# r03, before the statistics moved from __shfl_xor (ds_bpermute) to DPP adds: tree 312.1 ms per step, no copy 306.3, no statistics 302.4,
# neither 300.9 -- the six shuffles per strip cost more than the 4-byte copy; after: 301.5 ms with everything in place.
set -e
CS=clip_decontamination_amd/csrc
if [ "$1" = build ]; then
  for n in 1 2 3; do
    ( hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -w -DSG_H2_PROD_ABL=$n -c $CS/gemm_bf16.hip -o $CS/_obj/gemm_bf16_pabl$n.o &&
      hipcc -shared -fPIC --offload-arch=gfx950 $(ls $CS/_obj/*.o | grep -v gemm_bf16) $CS/_obj/gemm_bf16_pabl$n.o -o clip_decontamination_amd/libsegearth_hip_pabl$n.so ) &
  done
  wait
  ls -la clip_decontamination_amd/libsegearth_hip_pabl*.so
else
  for which in tree 1 2 3 tree; do
    if [ $which = tree ]; then unset SEGEARTH_HIP_LIB; else export SEGEARTH_HIP_LIB=$PWD/clip_decontamination_amd/libsegearth_hip_pabl$which.so; fi
    timeout -k 10 300 python bench.py --precision f16x2 --steps 3 --warmup 1 --no-cpu-baseline --no-self-check > gpurun_out/ab_h2p_$which.log 2>&1
    echo "$which:" $(tail -1 gpurun_out/ab_h2p_$which.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])")
  done
fi
