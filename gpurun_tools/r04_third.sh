#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for args in "" "--no-loss-chase" "" "--no-loss-chase"; do
  timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline $args 2>>gpurun_out/r04c_ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', d['ms_per_step'], d['config']['custom_loss'][:40], d['roofline']['avg_us'], d['config']['final_loss'])" | tee -a gpurun_out/r04c_ab.log
done
G2V_BENCH_ARGS="" bash gpurun_tools/prof_step.sh r04c | head -8
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04c_all_tests.log
cat gpurun_out/r04c_all_tests.log
