#!/bin/bash
# A/B: the step's parallel branches below 1024 rows per batch (off by default until round 5), several shapes
for cfg in "--config native" "--config genea --batch 128" "--batch 128" "--batch 512" "--config native --batch 512"; do
for rep in 1 2; do
for v in 1024 0; do
  r=$(G2V_OVERLAP_MIN_ROWS=$v timeout 300 python bench.py $cfg --steps 300 --warmup 10 --no-cpu-baseline --no-part-d --sustained 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$cfg MIN_ROWS=$v $r"
done
done
done
