#!/bin/bash
# Per-step kernel table of the CAPTURED pre-training step only (tools/soak_train_graph.py N replays; the 6 eager record /
# warm-up steps are in the trace too: N = 60 keeps their share under 10 %).   usage: bash tools/train_graph_kernels.sh [N]
R=$GRAFT_REPO_ROOT; N=${1:-60}; OUT=$R/gpurun_out/train_graph_kernels; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python tools/soak_train_graph.py $N > $OUT/run.log 2> $OUT/err.log)
cd $R
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)[0])))
n = $N + 6
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernel time per step %.2f ms, launches per step %.0f" % (tot / n / 1e6, calls / n))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%-100s calls/step %6.1f  ms/step %6.3f" % (r["Name"][:100], int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e6))
PY
tail -2 $OUT/run.log
rm -rf $OUT/kt
