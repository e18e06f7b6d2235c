#!/bin/bash
# after the last generic-kernel changes (precise wait counts, branch join): default line + the generic-shape lines + their kernel statistics
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r04_g_bench_default.json 2> gpurun_out/r04_g_bench_default.err
python - <<P
import json
d = json.loads(open("gpurun_out/r04_g_bench_default.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["frac"], [(r["att"], r["B"], r["ms_per_step"]) for r in d["text2embedding"]["runs"]])
P
: > gpurun_out/r04_g_bench_variants.jsonl
for args in "--steps 300 --warmup 10" "--batch 8192 --steps 50" "--batch 4100 --steps 50" "--config native --steps 200" "--config native --batch 4096 --steps 50" "--config genea --steps 50" "--config genea --batch 128 --steps 200"; do
  timeout 300 python bench.py --no-cpu-baseline --no-part-d $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'; keep['rollout'] = d['config']['decoder_rollout'][:40]; keep['whole_step_frac'] = d['roofline']['whole_step']['frac']
print(json.dumps(keep))" | tee -a gpurun_out/r04_g_bench_variants.jsonl
done
bash gpurun_tools/r04_prof_cfg.sh native 4096 | head -10; mv gpurun_out/r04_kernel_stats_native_B4096.csv gpurun_out/r04_g_kernel_stats_native_B4096.csv
bash gpurun_tools/r04_tl_cfg.sh native 4096 > /dev/null; cp gpurun_out/r04_timeline_native_B4096_libg2v_hip.txt gpurun_out/r04_g_step_timeline_native_B4096.txt
bash gpurun_tools/r04_prof_t2e.sh 4096 False | head -8; mv gpurun_out/r04_e_kernel_stats_part_d_B4096_attFalse.csv gpurun_out/r04_g_kernel_stats_part_d_B4096_noatt.csv
