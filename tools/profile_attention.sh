#!/bin/bash
# SQ counters of the attention kernel at the ViT-L/14 tile shape (each --pmc set in its OWN run, --kernel-trace only);
# outputs under gpurun_out/prof_attn_<tag>/.  profiles/r02_attention_sq_counters.json was condensed from such a run.
set -e
TAG=${1:-x}
OUT=$PWD/gpurun_out/prof_attn_$TAG
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
python3 $ROOT/tools/bench_attention.py > $OUT/time.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS"; do
  ATTN_REPS=2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 $ROOT/tools/bench_attention.py > $OUT/p$i.log 2> $OUT/p$i.err || echo "set $i failed" >> $OUT/time.log
  rm -f $OUT/p$i/c_kernel_trace.csv
  i=$((i+1))
done
cat $OUT/time.log
