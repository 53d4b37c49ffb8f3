#!/bin/bash
# rocprofv3 kernel trace of the headline leg at a given memory depth -> durations of the aggregation kernels.
# usage: bash tools/agg_trace.sh <mem_steps> [extra bench flags]
R=$GRAFT_REPO_ROOT; SRC=${AGG_SRC:-$R}; T=${1:-1}; shift; OUT=$R/gpurun_out/aggtrace_t$T; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $SRC/bench.py --mem-steps $T --no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-producer-leg --no-train-leg --no-roofline --steps 10 "$@" > $OUT/bench.json 2> $OUT/err.log
cd $R
python - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)[0])):
    if "grid_aggregate" in r["Name"] or "grid_relevance" in r["Name"]:
        print(r["Name"][:70], "calls", r["Calls"], "avg_us %.1f min %.1f max %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("t=$T", round(d["value"]), round(d["ms_per_step"],3))
PY
rm -rf $OUT/kt
