#!/bin/bash
# L2 counters of the GEMM launches of the default bench step (eager launches), grouped by grid size: hit rate of the
# operand stream (TCC_HIT / TCC_MISS), L2 read requests and what the fabric side served.  GPU box, repo root:
#   bash tools/pmc_step_gemm_l2.sh
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_step_gemm_l2; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum \
  --kernel-trace --kernel-include-regex "linear_planes" --output-format csv -d $OUT -o g -- \
  python $REPO/bench.py --eager --steps 4 --warmup 1 --no-roofline --no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-train-leg --no-producer-leg > $OUT/run.log 2>&1
cd $REPO
python tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv" | head -1) linear_planes > $OUT/summary.txt
cat $OUT/summary.txt
