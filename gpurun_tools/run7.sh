#!/bin/bash
mkdir -p gpurun_out/r2g
P=./gpurun_tools/px_test
for args in "256 33 0 0" "256 33 0 1" "256 33 168 1" "2 33 0 0" "20 33 50 1" "256 200 100 1"; do timeout 60 $P $args >> gpurun_out/r2g/px.txt 2>&1; done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_decoder_step.py -m gpu -q --tb=short -k "dec_rollout or decoder_step" 2>&1 | grep -v "^E    .*where" | tail -60 > gpurun_out/r2g/pytest_dec.txt
timeout 300 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2g/bench_persist.json 2> gpurun_out/r2g/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > /tmp/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2g/kernel_stats_persist.csv
cat gpurun_out/r2g/px.txt; tail -50 gpurun_out/r2g/pytest_dec.txt
for f in gpurun_out/r2g/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['roofline']['avg_us'])"; done
head -4 gpurun_out/r2g/kernel_stats_persist.csv
