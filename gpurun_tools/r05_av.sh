#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "compose2 or linear" 2>&1 | tail -3
timeout 2000 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_loss_chase.py tests/test_gpu_dp_engine.py tests/test_gpu_shipped_gssoft.py tests/test_gpu_train_script.py -q -x 2>&1 | tail -4
: > gpurun_out/r05_av_bench_variants.jsonl
for c in 1 0; do export G2V_COMPOSE_IN=$c; for args in "--config native --steps 300" "--config native --batch 4096 --steps 50" "--config genea --steps 50" "--config genea --batch 128 --steps 300"; do
  timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'; keep['compose'] = $c; keep['whole_step_frac'] = d['roofline']['whole_step']['frac']
print(json.dumps(keep))" | tee -a gpurun_out/r05_av_bench_variants.jsonl
done; done
