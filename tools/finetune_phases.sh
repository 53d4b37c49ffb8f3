#!/bin/bash
# Where a fine-tune iteration spends its wall time: kernel trace of tools/bench_finetune.py, last iteration cut at the
# optimizer launch (multi_adamw_kernel) -- per 5 ms window: GPU-busy fraction and launches; forward / backward split at the
# first backward kernel (the first linear_planes_tn_kernel of the iteration).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/finetune_phases; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && PYTHONPATH=. rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python tools/bench_finetune.py "$@" > $OUT/run.log 2> $OUT/err.log)
cd $R
python - <<PY
import csv, glob, collections
rows = sorted(csv.DictReader(open(glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "multi_adamw_kernel" in r["Kernel_Name"]]
a, b = marks[-2] + 1, marks[-1] + 1
it = rows[a:b]
t0 = int(it[0]["Start_Timestamp"]); t1 = int(it[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it)
print("last iteration: %d launches, span %.2f ms, kernel time %.2f ms (busy %.2f)" % (len(it), (t1 - t0) / 1e6, busy / 1e6, busy / (t1 - t0)))
fb = next(i for i, r in enumerate(it) if "linear_planes_tn_kernel" in r["Kernel_Name"])
tb = int(it[fb]["Start_Timestamp"])
for name, part, s, e in (("forward (rollout)", it[:fb], t0, tb), ("backward + optimizer", it[fb:], tb, t1)):
    kb = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part)
    print("%-22s %5d launches, span %7.2f ms, kernel time %7.2f ms (busy %.2f), %.1f us of span per launch" % (name, len(part), (e - s) / 1e6, kb / 1e6, kb / max(e - s, 1), (e - s) / 1e3 / max(len(part), 1)))
W = 5e6
win = collections.defaultdict(lambda: [0, 0])
for r in it:
    w = int((int(r["Start_Timestamp"]) - t0) // W)
    win[w][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); win[w][1] += 1
print("window (5 ms): busy fraction / launches")
print(" ".join("%d:%.2f/%d" % (w, win[w][0] / W, win[w][1]) for w in sorted(win)))
for name, part in (("forward", it[:fb]), ("backward", it[fb:])):
    c = collections.defaultdict(lambda: [0, 0])
    for r in part:
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:70]
        c[n][0] += 1; c[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("-- %s: top kernels" % name)
    for n, v in sorted(c.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %-72s %5d  %7.3f ms" % (n, v[0], v[1] / 1e6))
print(open("$OUT/run.log").read().strip().splitlines()[-1][:200])
PY
rm -rf $OUT/kt
