#!/bin/bash
# round 4, first GPU pass: the loss chaser's parity tests, the suites that touch what this round changed, a same-box A/B of the
# bench line (chaser on / off) and the one-step timeline of the default step
set -o pipefail
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_loss_chase.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04a_chase_tests.log
cat gpurun_out/r04a_chase_tests.log
for args in "" "--no-loss-chase" "" "--no-loss-chase"; do
  timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline $args 2>>gpurun_out/r04a_ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', d['ms_per_step'], d['config']['custom_loss'][:40], d['roofline']['avg_us'], d['config']['final_loss'])" | tee -a gpurun_out/r04a_ab.log
done
G2V_BENCH_ARGS="" bash gpurun_tools/prof_step.sh r04a
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04a_all_tests.log
cat gpurun_out/r04a_all_tests.log
