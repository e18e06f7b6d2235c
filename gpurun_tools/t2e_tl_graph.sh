#!/bin/bash
# Part d's iteration replayed from its hipGraph at batch $1, attention $2, side branches $3 (1/0): kernel timeline anchored at clip+Adam
B=${1:-128}; att=${2:-False}; side=${3:-1}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_t2etlg
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t2etlg -- python3 gpurun_tools/prof_t2e_graph.py $B $att $side > gpurun_out/prof_t2etlg.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_t2etlg/*/*kernel_trace.csv | head -1)
python gpurun_tools/timeline.py $f clip_adam | sed -n '/^ *0.0 dur/,$p' | cut -c1-130 > gpurun_out/t2e_graph_timeline_B${B}_att${att}_side${side}.txt
rm -rf gpurun_out/prof_t2etlg
