#!/bin/bash
mkdir -p gpurun_out/r2c
python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_dp_engine.py tests/test_gpu_ops.py -m gpu -q 2>&1 | tail -8 > gpurun_out/r2c/pytest.txt
python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2c/bench_wt.json 2> gpurun_out/r2c/err.txt
G2V_PLAIN_STORES=1 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2c/bench_plain.json 2>> gpurun_out/r2c/err.txt
python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2c/bench_wt2.json 2>> gpurun_out/r2c/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2c/prof_wt -o wt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r2c/prof_wt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2c/kernel_stats_wt.csv
rm -rf gpurun_out/r2c/prof_wt
cat gpurun_out/r2c/pytest.txt; for f in gpurun_out/r2c/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['roofline']['avg_us'])"; done
head -12 gpurun_out/r2c/kernel_stats_wt.csv
