#!/bin/bash
mkdir -p gpurun_out/r2f
P=./gpurun_tools/px_test
for args in "256 33 0 0" "256 33 0 1" "256 33 168 1" "2 33 0 0" "20 33 50 1" "256 200 100 1"; do timeout 60 $P $args >> gpurun_out/r2f/px.txt 2>&1; done
timeout 120 python gpurun_tools/persist_diag.py 32 6 > gpurun_out/r2f/diag32.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -k "dec_rollout" 2>&1 | tail -40 > gpurun_out/r2f/pytest_dec.txt
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_dp_engine.py -m gpu -q --tb=short 2>&1 | grep -v "^E    .*where" | tail -60 > gpurun_out/r2f/pytest_fail.txt
timeout 300 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2f/bench_persist.json 2> gpurun_out/r2f/err.txt
G2V_NO_PERSIST=1 timeout 300 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2f/bench_nopersist.json 2>> gpurun_out/r2f/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > /tmp/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2f/kernel_stats_persist.csv
cat gpurun_out/r2f/px.txt gpurun_out/r2f/diag32.txt; tail -25 gpurun_out/r2f/pytest_dec.txt; tail -40 gpurun_out/r2f/pytest_fail.txt
for f in gpurun_out/r2f/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['roofline']['avg_us'])"; done
head -14 gpurun_out/r2f/kernel_stats_persist.csv
