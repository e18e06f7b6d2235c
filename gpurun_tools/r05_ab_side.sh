#!/bin/bash
# round 5: A/B on one box of the step's side branch (forked in front of the input layer, masks last; merged prepare launch)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_loss_chase.py tests/test_gpu_train_script.py tests/test_gpu_dp_engine.py -x -q 2>&1 | tail -4
: > gpurun_out/r05_b_side_ab.log
for rep in 1 2 3; do for cfg in "0 0" "0 1" "1 0" "1 1"; do
  set -- $cfg
  echo -n "SIDE_EARLY=$1 MERGED_PREPARE=$2 " | tee -a gpurun_out/r05_b_side_ab.log
  G2V_SIDE_EARLY=$1 G2V_MERGED_PREPARE=$2 timeout 300 python gpurun_tools/bench_attr.py --steps 300 --warmup 10 --no-cpu-baseline --no-part-d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'], d['roofline'].get('avg_us'))" | tee -a gpurun_out/r05_b_side_ab.log
done; done
G2V_BENCH_ARGS="--no-part-d" bash gpurun_tools/prof_step.sh r05_b > /dev/null
cp gpurun_out/step_timeline_r05_b.txt gpurun_out/r05_b_step_timeline.txt
sed -n '/^ *0.0 dur/,$p' gpurun_out/r05_b_step_timeline.txt | cut -c1-118
