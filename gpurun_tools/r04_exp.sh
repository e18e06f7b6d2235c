#!/bin/bash
for dbg in 0 256 512 2 0 256 512 2; do
  G2V_DBG=$dbg timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dbg $dbg', d['ms_per_step'], d['config']['final_loss'])"
done
