#!/bin/bash
# A/B of one environment switch on the same box: bash gpurun_tools/ab_env.sh VAR valA valB  (alternating 200-step lines)
for rep in 1 2 3; do for v in "$2" "$3"; do
  echo -n "$1=$v "; env "$1=$v" timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
done; done
