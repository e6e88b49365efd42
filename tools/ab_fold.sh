#!/bin/bash
# same-box A/B of the folded LayerNorm (tuning -1 = default) against the separate pass (tuning 34); prints Mpix/s and the mean persistent-GEMM launch
for i in 1 2; do for t in 34 -1; do
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-self-check --tuning $t > gpurun_out/ab_$t.log 2>&1
  echo "tuning $t:" $(tail -1 gpurun_out/ab_$t.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])")
done; done
