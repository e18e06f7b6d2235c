#!/bin/bash
# A/B at the native shape (B = 128): the step's parallel branches (off below 1024 rows by default) now that the step is 54 launches
for rep in 1 2; do
for v in "15 1024" "4 0" "6 0" "15 0"; do
  set -- $v
  r=$(G2V_OVERLAP=$1 G2V_OVERLAP_MIN_ROWS=$2 timeout 300 python bench.py --config native --steps 300 --warmup 10 --no-cpu-baseline --no-part-d --sustained 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "G2V_OVERLAP=$1 MIN_ROWS=$2 $r"
done
done
