#!/bin/bash
# round-2 final measurements: default bench line, kernel stats + one-step timeline of the same command, Part d, variants
python bench.py > gpurun_out/r02_c_bench_default.json 2> gpurun_out/r02_c_bench_default.err
tail -c 600 gpurun_out/r02_c_bench_default.json
for v in "--steps 200 --warmup 10 --no-cpu-baseline" "--no-graph --no-cpu-baseline" "--batch 128 --no-cpu-baseline" "--dropout 0.2 --no-cpu-baseline"; do
  echo "== bench.py $v"; python bench.py $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'args': '$v', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'launch': d['config']['launch']}))" | tee -a gpurun_out/r02_c_bench_variants.jsonl
done
for m in 0 7; do echo "== G2V_OVERLAP=$m"; G2V_OVERLAP=$m python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'G2V_OVERLAP': $m, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))" | tee -a gpurun_out/r02_c_bench_variants.jsonl; done
python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 > gpurun_out/r02_c_part_d_bench.json; cat gpurun_out/r02_c_part_d_bench.json
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2d -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r2d.log 2>&1
f=$(ls gpurun_out/prof_r2d/*/*kernel_trace.csv | head -1); python gpurun_tools/timeline.py $f > gpurun_out/r02_c_step_timeline.txt; tail -3 gpurun_out/r02_c_step_timeline.txt
cp $(ls gpurun_out/prof_r2d/*/*kernel_stats.csv | head -1) gpurun_out/r02_c_kernel_stats_bench_steps30.csv
