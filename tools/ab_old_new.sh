#!/bin/bash
# same-box A/B: the committed tree exported to _ab_old/ (built there) against the working tree; bench legs alternate.
R=$GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-torch-gpu-baseline --no-depth-legs --no-producer-leg --no-train-leg --no-roofline --steps 40 --warmup 5"
for rep in 1 2; do
for cfg in "--mem-steps 1" "--mem-steps 5" "--shape native --mem-steps 1" "--shape native --mem-steps 5" "--shape native --mem-steps 15"; do
  for side in old new; do
    if [ $side = old ]; then D=$R/_ab_old; else D=$R; fi
    (cd $D && python bench.py $F $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$side', '$cfg', round(d['ms_per_step'],4))")
  done
done
done
