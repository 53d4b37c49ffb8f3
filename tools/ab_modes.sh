#!/bin/bash
R=$GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-producer-leg --no-train-leg --no-roofline --steps 40 --warmup 5"
for rep in 1 2 3; do
  (cd $R/_ab_old && python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', round(d['ms_per_step'],4))")
  for m in 0 1 2; do
    (cd $R && GRIDMM_AGG_MODE=$m python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode $m', round(d['ms_per_step'],4), d.get('replay_check'))")
  done
done
