#!/bin/bash
# SQ counters + effective clock of the persistent GEMM on the four ViT-L/14 linear shapes (tools/bench_gemm.py 30, plain epilogues);
# each --pmc set in its OWN run, --kernel-trace only.  Outputs under gpurun_out/prof_gemm_<tag>/.
set -e
TAG=${1:-x}
OUT=$PWD/gpurun_out/prof_gemm_$TAG
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  GEMM_TILES=128 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 $ROOT/tools/bench_gemm.py 30 > $OUT/p$i.log 2> $OUT/p$i.err || echo "set $i failed"
  i=$((i+1))
done
GEMM_TILES=128 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c -- python3 $ROOT/tools/bench_gemm.py 30 > $OUT/stats.log 2> $OUT/stats.err
grep "cfg 30" $OUT/stats.log
