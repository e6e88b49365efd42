#!/bin/bash
# Builds four ablated copies of the library (SG_PS_ABL = 1..4 in gemm_bf16.hip: the persistent GEMM without its steady-loop LDS-DMA issue /
# fragment reads / MFMAs / epilogue -- WRONG results by design) next to the in-tree one, for tools/bench_gemm.py:
#   tools/ablate_persist.sh build            (in the build container)
#   tools/ablate_persist.sh run              (on the GPU box: one process per library, same box, GEMM_TILES tiles)
set -e
CS=clip_decontamination_amd/csrc
if [ "$1" = build ]; then
  for n in ${ABLS:-1 2 3 4 5}; do
    ( hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -w -DSG_PS_ABL=$n -c $CS/gemm_bf16.hip -o $CS/_obj/gemm_bf16_abl$n.o &&
      hipcc -shared -fPIC --offload-arch=gfx950 $(ls $CS/_obj/*.o | grep -v gemm_bf16) $CS/_obj/gemm_bf16_abl$n.o -o clip_decontamination_amd/libsegearth_hip_abl$n.so ) &
  done
  wait
  ls -la clip_decontamination_amd/libsegearth_hip_abl*.so
else
  export GEMM_NOCHECK=1 GEMM_TILES=${GEMM_TILES:-119}
  for which in tree ${ABLS:-1 2 3 4 5}; do
    if [ $which = tree ]; then unset SEGEARTH_HIP_LIB; else export SEGEARTH_HIP_LIB=$PWD/clip_decontamination_amd/libsegearth_hip_abl$which.so; fi
    echo "== ablation $which"
    timeout -k 10 200 python tools/bench_gemm.py 30 2>&1 | grep TFLOP
  done
  unset SEGEARTH_HIP_LIB
  for cap in 256 128 64 32; do echo "== persistent grid capped at $cap workgroups"; GEMM_GRID_CAP=$cap timeout -k 10 200 python tools/bench_gemm.py 30 2>&1 | grep TFLOP; done
fi
