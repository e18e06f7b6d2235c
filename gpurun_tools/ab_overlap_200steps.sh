#!/bin/bash
# A/B of two G2V_OVERLAP masks, alternated, 200 timed steps each
for rep in 1 2 3 4; do
  for m in 7 15; do
    G2V_OVERLAP=$m python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['ms_per_step'])"
  done
done
