#!/bin/bash
# SQ counters of the attention kernels over tools/bench_attn2.py (GPU box).  usage: bash tools/pmc_attn.sh
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_attn; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace --kernel-include-regex "attention_rows" --output-format csv -d $OUT -o a -- python $REPO/tools/bench_attn2.py ${1:-2} > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM \
  --kernel-trace --kernel-include-regex "attention_rows" --output-format csv -d $OUT -o b -- python $REPO/tools/bench_attn2.py ${1:-2} >> $OUT/run.log 2>&1
cd $REPO
for f in $(find $OUT -name "*counter_collection.csv"); do python tools/pmc_summary.py $f attention_rows; done > $OUT/summary.txt
head -c 8000 $OUT/summary.txt
