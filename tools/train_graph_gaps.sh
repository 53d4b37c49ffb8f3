#!/bin/bash
# Is the replay of the captured training step GPU-bound or dispatch-bound?  Kernel trace of tools/dbg_graph_train3.py
# (full-size, one task), then busy time vs span of the last replays.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/train_gaps; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 TIMING=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python tools/dbg_graph_train3.py full ${1:-mlm} > $OUT/run.log 2>&1)
cd $R; tail -2 $OUT/run.log
python - <<PY
import csv, glob
rows = sorted(csv.DictReader(open(glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
# the last 10 replays are back to back: take the final 40 % of the kernels and measure busy / span
tail = rows[int(len(rows) * 0.6):]
s0, e1 = int(tail[0]["Start_Timestamp"]), int(tail[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(tail, tail[1:])]
big = sorted(gaps, reverse=True)[:5]
print("kernels", len(tail), "span %.1f ms busy %.1f ms (%.0f %%)" % ((e1 - s0) / 1e6, busy / 1e6, 100.0 * busy / (e1 - s0)), "median gap %.2f us" % (sorted(gaps)[len(gaps) // 2] / 1e3), "largest gaps us", [g / 1e3 for g in big])
PY
rm -rf $OUT/kt
