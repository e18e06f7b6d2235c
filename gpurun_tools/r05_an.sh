#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "fold2 or small_row or linear" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_vqvae.py -q -x 2>&1 | tail -3
: > gpurun_out/r05_an_bench_variants.jsonl
for args in "--config native --steps 300" "--config native --steps 300" "--config native --batch 4096 --steps 50" "--config genea --steps 50" "--config genea --batch 128 --steps 300"; do
  timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'; keep['whole_step_frac'] = d['roofline']['whole_step']['frac']
print(json.dumps(keep))" | tee -a gpurun_out/r05_an_bench_variants.jsonl
done
bash gpurun_tools/r04_tl_cfg.sh native 128 > /dev/null 2>&1; cp gpurun_out/r04_timeline_native_B128_libg2v_hip.txt gpurun_out/r05_an_timeline_native_B128.txt; sed -n '/dec_cluster_bwd/,$p' gpurun_out/r05_an_timeline_native_B128.txt | cut -c1-130
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | cut -c1-200
