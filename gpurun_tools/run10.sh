#!/bin/bash
mkdir -p gpurun_out/r2j
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E    .*where" | tail -60 > gpurun_out/r2j/pytest.txt
timeout 300 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > /tmp/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2j/kernel_stats.csv
tail -40 gpurun_out/r2j/pytest.txt
python -c "import json; d=json.load(open('gpurun_out/r2j/bench.json')); print(d['ms_per_step'], d['roofline'])"
head -30 gpurun_out/r2j/kernel_stats.csv | cut -c1-150
