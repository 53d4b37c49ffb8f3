#!/bin/bash
# rocprofv3 kernel trace of bench.py (headline leg only) -> per-(kernel, grid) table.  usage: bash tools/trace_step.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-r2}; OUT=$R/gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-producer-leg --no-train-leg --no-roofline --steps 20 > $OUT/bench.json 2> $OUT/err.log
cd $R
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python tools/trace_summary.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) auto 60 > $OUT/trace_by_shape.txt 2>&1
rm -rf $OUT/kt
