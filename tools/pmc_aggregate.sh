#!/bin/bash
# SQ counters of the aggregation kernels over the default bench workload (eager launches).  GPU box, repo root:
#   bash tools/pmc_aggregate.sh [extra bench.py flags, e.g. --shape native --mem-steps 5]
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_agg; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --kernel-include-regex "grid_aggregate|grid_relevance" --output-format csv -d $OUT -o a -- \
  python $REPO/bench.py --eager --steps 4 --warmup 1 --no-roofline --no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-train-leg --no-producer-leg "$@" > $OUT/run.log 2>&1
cd $REPO
python tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv" | head -1) "grid_aggregate|grid_relevance" > $OUT/summary.txt
cat $OUT/summary.txt
