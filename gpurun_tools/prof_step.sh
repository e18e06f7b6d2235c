#!/bin/bash
# one-step timeline of the default bench step -> gpurun_out/step_timeline_$1.txt  (environment passes through)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_s$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_s$tag -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline $G2V_BENCH_ARGS > gpurun_out/prof_s$tag.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_s$tag/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python gpurun_tools/timeline.py "$f" > gpurun_out/step_timeline_$tag.txt; fi
rm -rf gpurun_out/prof_s$tag
sed -n '/^ *0.0 dur/,$p' gpurun_out/step_timeline_$tag.txt | cut -c1-118
