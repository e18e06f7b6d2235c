#!/bin/bash
# A/B of the product library against gpurun_tools/libg2v_alt.so on one box (alternating 200-step lines)
for rep in 1 2 3; do for l in gesture2vec_amd/libg2v_hip.so gpurun_tools/libg2v_alt.so; do
  echo -n "$l "; timeout 300 python gpurun_tools/bench_altlib.py $l --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
done; done
