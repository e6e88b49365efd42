#!/bin/bash
# rocprofv3 evidence for the non-headline modes (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/):
#   fp8:  timing pass of `bench.py --precision fp8`
#   jbu:  timing pass + FETCH_SIZE / WRITE_SIZE passes (each in its OWN run, --kernel-trace only) of tools/bench_jbu.py ViT-B/16
# Afterwards, locally:  python tools/pmc_summary.py <stats.csv> <fetch.csv> <write.csv> profiles/<name>.json
set -e
TAG=${1:-round}
OUT=$PWD/gpurun_out/prof_$TAG
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fp8 -o b -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-self-check --precision fp8 > $OUT/bench_fp8.json 2> $OUT/fp8.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/jbu -o j -- python3 $ROOT/tools/bench_jbu.py ViT-B/16 > $OUT/jbu.log 2> $OUT/jbu.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/jbu_fetch -o f -- python3 $ROOT/tools/bench_jbu.py ViT-B/16 > $OUT/jbu_fetch.log 2> $OUT/jbu_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/jbu_write -o w -- python3 $ROOT/tools/bench_jbu.py ViT-B/16 > $OUT/jbu_write.log 2> $OUT/jbu_write.err
rm -f $OUT/fp8/b_kernel_trace.csv $OUT/jbu/j_kernel_trace.csv $OUT/jbu_fetch/f_kernel_trace.csv $OUT/jbu_write/w_kernel_trace.csv
ls $OUT
