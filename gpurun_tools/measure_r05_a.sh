#!/bin/bash
# round 5, start of round: the "before" numbers on one box
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r05_a_bench_default.json 2> gpurun_out/r05_a_bench_default.err
tail -c 3000 gpurun_out/r05_a_bench_default.json
for args in "--steps 300 --warmup 10" "--batch 128 --steps 300" "--config native --steps 200" "--config native --batch 4096 --steps 50" "--config genea --steps 50"; do
  timeout 300 python bench.py --no-cpu-baseline --no-part-d $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'
print(json.dumps(keep))" | tee -a gpurun_out/r05_a_bench_variants.jsonl
done
bash gpurun_tools/r05_pmc_rollout.sh a | tail -80
