#!/bin/bash
# Two separate PMC passes over the default bench workload (eager launches, no baselines), then the per-launch summary.
# Run on the GPU box from the repo root:  bash tools/collect_traffic.sh r2
set -e
TAG=${1:-r2}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --eager --steps 4 --warmup 1 --no-roofline --no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-train-leg --no-producer-leg"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
cd $REPO
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python tools/hbm_traffic.py $F $W $OUT/hbm_traffic.json batch=32 shape=baseline mem_steps=1 | tee $OUT/hbm_traffic.txt
