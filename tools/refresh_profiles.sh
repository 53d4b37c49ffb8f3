set -x
# Regenerates the round's evidence on a GPU box: bench line, HBM-traffic PMC passes, kernel stats + per-shape trace.
# usage (from the repo root on the box): bash tools/refresh_profiles.sh r2     -> gpurun_out/refresh/
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; TAG=${1:-r2}
mkdir -p gpurun_out/refresh
bash tools/collect_traffic.sh $TAG > gpurun_out/refresh/traffic.log 2>&1
cp gpurun_out/pmc_$TAG/hbm_traffic.json gpurun_out/refresh/${TAG}_hbm_traffic.json; cp gpurun_out/pmc_$TAG/hbm_traffic.txt gpurun_out/refresh/${TAG}_hbm_traffic.txt
cp gpurun_out/refresh/${TAG}_hbm_traffic.json profiles/${TAG}_hbm_traffic.json     # bench.py reads the newest one
timeout 1200 python bench.py > gpurun_out/refresh/${TAG}_bench_n1.json 2> gpurun_out/refresh/bench_n1.err
tail -c 400 gpurun_out/refresh/${TAG}_bench_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/refresh/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-train-leg --no-producer-leg --no-roofline > $R/gpurun_out/refresh/kt.log 2>&1
cd $R
S=$(find gpurun_out/refresh/kt -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/refresh/${TAG}_bench_kernel_stats.csv
T=$(find gpurun_out/refresh/kt -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $T auto 60 > gpurun_out/refresh/${TAG}_bench_trace_by_shape.txt 2>&1
head -5 gpurun_out/refresh/${TAG}_bench_trace_by_shape.txt
# the per-step kernel sum of the trace must reproduce the bench line's step time (the evidence file checks itself)
python - <<PY
import json, re
d = json.loads(open("gpurun_out/refresh/${TAG}_bench_n1.json").read().strip().splitlines()[-1])
txt = open("gpurun_out/refresh/${TAG}_bench_trace_by_shape.txt").read()
tot = float(re.search(r"total per step us ([0-9.]+)", txt).group(1)) / 1e3
print("trace: %.3f ms of kernels per step; bench: %.3f ms per step" % (tot, d["ms_per_step"]))
assert abs(tot - d["ms_per_step"]) < 0.10 * d["ms_per_step"], "per-step kernel sum and ms_per_step disagree by more than 10 %"
PY
rm -rf gpurun_out/refresh/kt gpurun_out/pmc_$TAG/fetch gpurun_out/pmc_$TAG/write
for t in 1 5 15; do timeout 300 python bench.py --shape native --mem-steps $t --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-train-leg --no-producer-leg > gpurun_out/refresh/${TAG}_bench_native_t$t.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/refresh/*bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"]), round(d["ms_per_step"],3), round(d["kernels"]["grid_aggregate"]["avg_us"],1) if "kernels" in d else "")
    except Exception as e: print(f, "ERR", e)
PY
