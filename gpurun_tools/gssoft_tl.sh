#!/bin/bash
# one-step timeline of the as-shipped soft-quantiser train iteration (gpurun_tools/gssoft_bench.py) at batch $1
B=${1:-4096}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
export G2V_ONLY=1
rm -rf gpurun_out/prof_gst
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gst -- python3 gpurun_tools/gssoft_bench.py $B > gpurun_out/prof_gst.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_gst/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python gpurun_tools/timeline.py "$f" | sed -n '/^ *0.0 dur/,$p' | cut -c1-110 > gpurun_out/gssoft_tl_B$B.txt; fi
rm -rf gpurun_out/prof_gst
cat gpurun_out/gssoft_tl_B$B.txt | head -90; tail -1 gpurun_out/prof_gst.log | cut -c1-300
