#!/bin/bash
# The headline step with other LDS-ring depths for the small / mid GEMM tiles (GRIDMM_GEMM_CFG_SMALL / _MID override
# pick_cfg's 64x64 / 128x128 choices): in the step every weight arrives cold (HBM), unlike in tools/bench_gemm.py.
R=${GRAFT_REPO_ROOT:-.}; cd $R
run() { echo "== SMALL=$1 MID=$2"; GRIDMM_GEMM_CFG_SMALL=$1 GRIDMM_GEMM_CFG_MID=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline \
  --no-torch-gpu-baseline --no-roofline --no-depth-legs --no-train-leg --no-producer-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ms_per_step %.4f  value %.0f  replay %s' % (d['ms_per_step'], d['value'], d['replay_check']))"; }
run 0 0
run 45 0
run 46 0
run 47 0
run 49 0
run 0 48
run 46 48
run 0 0
