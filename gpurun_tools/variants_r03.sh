#!/bin/bash
# round-3 bench variants: batch sizes, dropout, eager launches; Part d
cd "${GRAFT_REPO_ROOT:?}"
rm -f gpurun_out/r03_d_bench_variants.jsonl
for v in "--batch 128" "--batch 1024" "--batch 2048" "--batch 4096 --steps 200 --warmup 10" "--batch 4100" "--batch 8192" "--dropout 0.2" "--no-graph"; do
  timeout 300 python bench.py $v --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'args': '$v', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'launch': d['config']['launch']}))" | tee -a gpurun_out/r03_d_bench_variants.jsonl
done
timeout 600 python gpurun_tools/bench_t2e.py 2>/dev/null < /dev/null | tail -1 > gpurun_out/r03_d_part_d_bench.json; cat gpurun_out/r03_d_part_d_bench.json
