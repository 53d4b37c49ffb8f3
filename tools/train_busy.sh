#!/bin/bash
# GPU-busy time of a pre-training step: kernel trace of tools/train_only.py (3 warm-up + N timed steps)
R=$GRAFT_REPO_ROOT; N=${1:-6}; OUT=$R/gpurun_out/train_busy; rm -rf $OUT; mkdir -p $OUT
cd $R && python tools/train_only.py $N > $OUT/plain.json 2>$OUT/plain.err; cat $OUT/plain.json
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python tools/train_only.py $N > $OUT/prof.json 2> $OUT/err.log)
cd $R
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
n = $N + 3
print("kernel time per step %.2f ms, launches per step %.0f" % (tot / n / 1e6, calls / n))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%-90s calls/step %6.1f  ms/step %6.3f" % (r["Name"][:90], int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e6))
PY
rm -rf $OUT/kt
