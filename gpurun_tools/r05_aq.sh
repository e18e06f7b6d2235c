#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "preclear or gru or dec_cluster or dec_rollout" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_loss_chase.py tests/test_gpu_dp_engine.py -q -x 2>&1 | tail -3
: > gpurun_out/r05_aq_bench_variants.jsonl
for pc in 1 0 1 0; do export G2V_XCH_PRECLEAR=$pc; for args in "--config native --steps 300" "--config genea --batch 128 --steps 300"; do
  timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'; keep['preclear'] = $pc
print(json.dumps(keep))" | tee -a gpurun_out/r05_aq_bench_variants.jsonl
done; done
export G2V_XCH_PRECLEAR=1
bash gpurun_tools/r04_tl_cfg.sh native 128 > /dev/null 2>&1; cp gpurun_out/r04_timeline_native_B128_libg2v_hip.txt gpurun_out/r05_aq_timeline_native_B128.txt; cut -c1-130 gpurun_out/r05_aq_timeline_native_B128.txt | tail -45
