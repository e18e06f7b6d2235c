#!/bin/bash
# generic-dims decoder step kernels with eight waves: parity tests + the native / genea bench lines
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "dec_rollout" -x 2>&1 | tail -8 > gpurun_out/r04_gen8_tests.log; cat gpurun_out/r04_gen8_tests.log
: > gpurun_out/r04_gen8_bench.jsonl
for args in "--config native --batch 4096 --steps 50" "--config genea --steps 50" "--config native --steps 200" "--config native --batch 1024 --steps 100" "--config native --batch 2048 --steps 100"; do
  timeout 300 python bench.py --no-cpu-baseline --no-part-d $args 2>gpurun_out/r04_gen8_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'; keep['whole_step_frac'] = d['roofline']['whole_step']['frac']
print(json.dumps(keep))" | tee -a gpurun_out/r04_gen8_bench.jsonl
done
tail -5 gpurun_out/r04_gen8_err.log
