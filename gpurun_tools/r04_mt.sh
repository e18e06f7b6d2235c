#!/bin/bash
# multi-tile persistent rollout: parity tests + bench at the batches beyond one tile per CU
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "multi_tile or persistent_matches" -x 2>&1 | tail -15 > gpurun_out/r04_mt_tests.log; cat gpurun_out/r04_mt_tests.log
: > gpurun_out/r04_mt_bench.jsonl
for args in "--batch 8192 --steps 50" "--batch 12288 --steps 30" "--batch 4112 --steps 50" "--batch 4096 --steps 100"; do
  timeout 300 python bench.py --no-cpu-baseline --no-part-d $args 2>gpurun_out/r04_mt_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'; keep['launch'] = d['config']['launch']
print(json.dumps(keep))" | tee -a gpurun_out/r04_mt_bench.jsonl
done
tail -5 gpurun_out/r04_mt_err.log
