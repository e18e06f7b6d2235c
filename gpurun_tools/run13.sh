#!/bin/bash
mkdir -p gpurun_out/r2m
timeout 900 python bench.py > gpurun_out/r2m/bench_default.json 2> gpurun_out/r2m/err.txt
timeout 300 python bench.py --dropout 0.2 --no-cpu-baseline > gpurun_out/r2m/bench_dropout02.json 2>> gpurun_out/r2m/err.txt
timeout 300 python bench.py --steps 200 --no-cpu-baseline > gpurun_out/r2m/bench_steps200.json 2>> gpurun_out/r2m/err.txt
timeout 300 python bench.py --force-dp --no-cpu-baseline > gpurun_out/r2m/bench_forcedp.json 2>> gpurun_out/r2m/err.txt
for b in 128 2048 4100 8192; do timeout 300 python bench.py --batch $b --no-cpu-baseline > gpurun_out/r2m/bench_B$b.json 2>> gpurun_out/r2m/err.txt; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > /tmp/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2m/kernel_stats_bench_steps30.csv
python gpurun_tools/vq_sweep.py > gpurun_out/r2m/vq_sweep.json 2>> gpurun_out/r2m/err.txt
for f in gpurun_out/r2m/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'], d['roofline']['avg_us'] if 'roofline' in d else '')"; done
cat gpurun_out/r2m/bench_default.json; cat gpurun_out/r2m/vq_sweep.json
