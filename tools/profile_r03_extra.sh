#!/bin/bash
# Round-3 evidence beyond the headline profile (tools/profile_round.sh): per-config bench lines, rocprofv3 summaries of BASELINE config 4 at its real
# shape (ViT-L/14 + jbu_one, C = 768) and of the exact (two-plane f16) mode, and the 2-rank rehearsals of the self-launching bench.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03x
mkdir -p $OUT
for c in 2 3 4 5; do timeout -k 10 400 python3 bench.py --config $c --steps 5 --warmup 2 > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err; done
timeout -k 10 400 python3 bench.py --precision f16x2 --steps 5 --warmup 2 > $OUT/bench_f16x2.json 2> $OUT/bench_f16x2.err
timeout -k 10 400 python3 bench.py --precision f16 --steps 5 --warmup 2 > $OUT/bench_f16.json 2> $OUT/bench_f16.err
timeout -k 10 400 python3 bench.py --precision fp8 --steps 5 --warmup 2 > $OUT/bench_fp8.json 2> $OUT/bench_fp8.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/jbu_stats -o jbu -- python3 bench.py --config 4 --steps 3 --warmup 1 --no-self-check > $OUT/jbu_stats.json 2> $OUT/jbu_stats.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/jbu_fetch -o f -- python3 bench.py --config 4 --steps 1 --warmup 1 --no-self-check > $OUT/jbu_fetch.json 2> $OUT/jbu_fetch.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/jbu_write -o w -- python3 bench.py --config 4 --steps 1 --warmup 1 --no-self-check > $OUT/jbu_write.json 2> $OUT/jbu_write.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h2_stats -o h2 -- python3 bench.py --precision f16x2 --steps 3 --warmup 1 --no-self-check --no-cpu-baseline > $OUT/h2_stats.json 2> $OUT/h2_stats.err
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --share-gpu --tile-rows 3 --steps 2 --warmup 1 > $OUT/bench_2rank_weak.json 2> $OUT/bench_2rank_weak.err
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --share-gpu --scaling strong --scene 3072 --steps 2 --warmup 1 > $OUT/bench_2rank_strong.json 2> $OUT/bench_2rank_strong.err
rm -f $OUT/*/*_kernel_trace.csv
ls $OUT
