#!/bin/bash
# same-box A/B of two builds of the library: the in-tree one against SEGEARTH_HIP_LIB=<other .so> (bench.py, no CPU baseline)
#   tools/ab_lib.sh [other.so] [bench.py options]
OTHER=${1:-clip_decontamination_amd/libsegearth_hip_old.so}
for i in 1 2 3; do
  for which in other tree; do
    if [ $which = other ]; then export SEGEARTH_HIP_LIB=$PWD/$OTHER; else unset SEGEARTH_HIP_LIB; fi
    timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-self-check ${@:2} > gpurun_out/ab_lib_$which.log 2>&1
    echo "$which:" $(tail -1 gpurun_out/ab_lib_$which.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'gemm', r['avg_launch_us'], 'attention', r['attention']['avg_launch_us'])")
  done
done
