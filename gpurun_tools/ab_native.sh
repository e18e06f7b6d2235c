#!/bin/bash
# A/B of the product library against gpurun_tools/libg2v_alt.so on one box at a config ($1, batch $2)
for rep in 1 2; do for l in gesture2vec_amd/libg2v_hip.so gpurun_tools/libg2v_alt.so; do
  echo -n "$l "; timeout 300 python gpurun_tools/bench_altlib.py $l --config ${1:-native} --batch ${2:-4096} --steps 50 --no-cpu-baseline --no-part-d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
done; done
