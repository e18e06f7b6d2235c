#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
export G2V_ONLY=${1:-1}
rm -rf gpurun_out/prof_gst
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gst -- python3 gpurun_tools/gssoft_bench.py 128 > gpurun_out/prof_gst.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_gst/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python gpurun_tools/timeline.py "$f" | sed -n '/^ *0.0 dur/,$p' | cut -c1-100 > gpurun_out/gssoft_tl_$G2V_ONLY.txt; fi
rm -rf gpurun_out/prof_gst
cat gpurun_out/gssoft_tl_$G2V_ONLY.txt | head -75
