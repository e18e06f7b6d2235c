#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "compose2" 2>&1 | tail -3
: > gpurun_out/r05_aw_bench_variants.jsonl
for c in 1 0 1 0; do export G2V_COMPOSE_IN=$c; for args in "--config native --steps 300" "--config genea --batch 128 --steps 300"; do
  timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'; keep['compose'] = $c
print(json.dumps(keep))" | tee -a gpurun_out/r05_aw_bench_variants.jsonl
done; done
export G2V_COMPOSE_IN=1
bash gpurun_tools/r04_tl_cfg.sh native 128 > /dev/null 2>&1; head -16 gpurun_out/r04_timeline_native_B128_libg2v_hip.txt | cut -c1-130
