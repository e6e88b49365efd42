#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun; outputs land in gpurun_out/prof_<tag>/).
#   timing pass:  --kernel-trace --stats            (the per-kernel averages bench.py's roofline must agree with)
#   PMC passes:   --pmc FETCH_SIZE / --pmc WRITE_SIZE, each in its OWN run with --kernel-trace only
# Afterwards, locally:  python tools/pmc_summary.py gpurun_out/prof_<tag>/stats/bench_kernel_stats.csv \
#                         gpurun_out/prof_<tag>/fetch/f_counter_collection.csv gpurun_out/prof_<tag>/write/w_counter_collection.csv profiles/<name>.json
set -e
TAG=${1:-round}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-self-check > $OUT/bench_stats.json 2> $OUT/stats.err
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-self-check > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-self-check > $OUT/bench_write.json 2> $OUT/write.err
rm -f $OUT/stats/bench_kernel_trace.csv $OUT/fetch/f_kernel_trace.csv $OUT/write/w_kernel_trace.csv
ls $OUT
