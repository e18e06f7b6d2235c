#!/bin/bash
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -k "dec_rollout" 2>&1 | tail -60 > gpurun_out/r2d/pytest_dec.txt
timeout 600 python -m pytest "tests/test_gpu_vqvae.py" tests/test_gpu_dp_engine.py -m gpu -q --tb=short 2>&1 | grep -v "^E    .*where" | tail -80 > gpurun_out/r2d/pytest_fail.txt
timeout 300 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2d/bench_persist.json 2> gpurun_out/r2d/err.txt
G2V_NO_PERSIST=1 timeout 300 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2d/bench_nopersist.json 2>> gpurun_out/r2d/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > /tmp/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2d/kernel_stats_persist_fwd.csv
tail -3 /tmp/prof_a.log > gpurun_out/r2d/prof_log.txt
cat gpurun_out/r2d/pytest_dec.txt | tail -30; cat gpurun_out/r2d/pytest_fail.txt | tail -60
for f in gpurun_out/r2d/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['roofline']['avg_us'])"; done
head -14 gpurun_out/r2d/kernel_stats_persist_fwd.csv
