for a in 0 1 2 4 7 3 5 6; do
  GRIDMM_AGG_ABLATE=$a python bench.py --no-cpu-baseline --no-torch-gpu-baseline --eager --steps 5 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('ablate', $a, round(d['kernels']['grid_aggregate']['avg_us'],1))"
done
