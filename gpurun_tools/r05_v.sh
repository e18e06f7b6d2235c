#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
for cfg in "" "--config native" "--config genea --batch 128" "--batch 128" "--batch 512" "--config native --batch 512" "--batch 1024"; do
  r=$(timeout 300 python bench.py $cfg --steps 300 --warmup 10 --no-cpu-baseline --no-part-d --sustained 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$cfg : $r"
done
