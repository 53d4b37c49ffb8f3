#!/bin/bash
# PMC pass over tools/bench_gemm.py for a few tile configs (GPU box).  usage: bash tools/pmc_gemm.sh "1 15 7"
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_gemm; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT -o g -- python $REPO/tools/bench_gemm.py $1 > $OUT/run.log 2>&1
cd $REPO
python tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv" | head -1) linear_planes > $OUT/summary.txt
head -c 6000 $OUT/summary.txt
