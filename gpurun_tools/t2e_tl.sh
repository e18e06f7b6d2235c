#!/bin/bash
# one Part-d train iteration (gpurun_tools/prof_t2e.py) at batch $1, attention $2, as a kernel timeline anchored at clip+Adam
B=${1:-4096}; att=${2:-False}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_t2etl
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t2etl -- python3 gpurun_tools/prof_t2e.py $B $att > gpurun_out/prof_t2etl.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_t2etl/*/*kernel_trace.csv | head -1)
python gpurun_tools/timeline.py $f clip_adam | sed -n '/^ *0.0 dur/,$p' | cut -c1-130 > gpurun_out/t2e_timeline_B${B}_att${att}.txt
rm -rf gpurun_out/prof_t2etl
