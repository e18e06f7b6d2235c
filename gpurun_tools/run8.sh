#!/bin/bash
mkdir -p gpurun_out/r2h
python gpurun_tools/pstamps.py > gpurun_out/r2h/pstamps.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -k "dec_rollout" 2>&1 | grep -v "^E    .*where" | tail -40 > gpurun_out/r2h/pytest_dec.txt
timeout 300 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2h/bench_persist.json 2> gpurun_out/r2h/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > /tmp/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2h/kernel_stats_persist.csv
tail -12 gpurun_out/r2h/pstamps.txt; tail -30 gpurun_out/r2h/pytest_dec.txt
for f in gpurun_out/r2h/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['roofline']['avg_us'])"; done
head -4 gpurun_out/r2h/kernel_stats_persist.csv
