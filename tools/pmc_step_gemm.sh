#!/bin/bash
# SQ counters of the GEMM launches of the default bench step (eager launches), grouped by grid size.  GPU box, repo root:
#   bash tools/pmc_step_gemm.sh
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_step_gemm; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --kernel-include-regex "linear_planes" --output-format csv -d $OUT -o g -- \
  python $REPO/bench.py --eager --steps 4 --warmup 1 --no-roofline --no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-train-leg --no-producer-leg > $OUT/run.log 2>&1
cd $REPO
python tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv" | head -1) linear_planes > $OUT/summary.txt
cat $OUT/summary.txt
