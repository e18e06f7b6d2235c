#!/bin/bash
# A/B: branch 4 (the encoder's GRU weight gradients beside the input layer's gradient chain), G2V_OVERLAP 15 vs 31
for cfg in "--config native" "--config genea --batch 128" "--batch 128" "" "--config native --batch 4096 --steps 50"; do
for rep in 1 2; do
for v in 15 31; do
  r=$(G2V_OVERLAP=$v timeout 300 python bench.py $cfg --steps 300 --warmup 10 --no-cpu-baseline --no-part-d --sustained 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$cfg OVERLAP=$v $r"
done
done
done
