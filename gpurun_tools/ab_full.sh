#!/bin/bash
# A/B of the product library against gpurun_tools/libg2v_alt.so on one box at the headline shape (300 steps, 3 rounds)
for rep in 1 2 3; do for l in gesture2vec_amd/libg2v_hip.so gpurun_tools/libg2v_alt.so; do
  echo -n "$l "; timeout 300 python gpurun_tools/bench_altlib.py $l --steps 300 --warmup 30 --no-cpu-baseline --no-part-d "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
done; done
