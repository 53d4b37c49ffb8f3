#!/bin/bash
# kernel trace of the headline leg -> one step's timeline (gaps between graph nodes).  usage: bash tools/timeline.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-r2}; OUT=$R/gpurun_out/timeline_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-producer-leg --no-train-leg --no-roofline --steps 10 > $OUT/bench.json 2> $OUT/err.log
cd $R
python tools/step_timeline.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 2 > $OUT/step_timeline.txt 2>&1
rm -rf $OUT/kt
tail -3 $OUT/step_timeline.txt
