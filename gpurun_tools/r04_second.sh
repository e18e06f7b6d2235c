#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout 300 python gpurun_tools/r04_dbg.py 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_loss_chase.py tests/test_gpu_dp_engine.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04b_chase_tests.log
cat gpurun_out/r04b_chase_tests.log
for args in "" "--no-loss-chase" "" "--no-loss-chase"; do
  timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline $args 2>>gpurun_out/r04b_ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', d['ms_per_step'], d['config']['custom_loss'][:40], d['roofline']['avg_us'], d['config']['final_loss'])" | tee -a gpurun_out/r04b_ab.log
done
G2V_BENCH_ARGS="" bash gpurun_tools/prof_step.sh r04b | head -8
