#!/bin/bash
# Per-iteration kernel table of the fine-tune leg (tools/bench_finetune.py: teacher-forced rollout + one backward + AdamW).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/finetune_kernels; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && PYTHONPATH=. rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python tools/bench_finetune.py > $OUT/run.log 2> $OUT/err.log)
cd $R
python - <<PY
import csv, glob, json
rows = list(csv.DictReader(open(glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)[0])))
line = [l for l in open("$OUT/run.log") if l.startswith("{")][-1]
n = json.loads(line)["iters"] + 2          # + warm-up iterations
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernel time per iteration %.2f ms, launches per iteration %.0f (n = %d iterations incl. warm-up)" % (tot / n / 1e6, calls / n, n))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%-100s calls/it %7.1f  ms/it %7.3f" % (r["Name"][:100], int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e6))
print(line.strip())
PY
rm -rf $OUT/kt
