#!/bin/bash
# round-3 measurements: default bench line (incl. cpu_baseline), kernel stats + one-step timeline of the same command
tag=${1:-a}
timeout 420 python bench.py > gpurun_out/r03_${tag}_bench_default.json 2> gpurun_out/r03_${tag}_bench_default.err
tail -c 400 gpurun_out/r03_${tag}_bench_default.json
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3${tag} -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r3${tag}.log 2>&1
f=$(ls gpurun_out/prof_r3${tag}/*/*kernel_trace.csv | head -1); python gpurun_tools/timeline.py $f > gpurun_out/r03_${tag}_step_timeline.txt; tail -3 gpurun_out/r03_${tag}_step_timeline.txt
cp $(ls gpurun_out/prof_r3${tag}/*/*kernel_stats.csv | head -1) gpurun_out/r03_${tag}_kernel_stats_bench_steps30.csv
rm -rf gpurun_out/prof_r3${tag}
