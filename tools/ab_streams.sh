#!/bin/bash
# same-box comparison of one HIP stream per rank (the default) against two (bench.py --streams 2: the halves' tails overlap)
for st in 1 2 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-self-check --streams $st > gpurun_out/ab_streams_$st.log 2>&1
  echo "streams $st:" $(tail -1 gpurun_out/ab_streams_$st.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])")
done
