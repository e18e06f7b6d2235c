#!/bin/bash
# one replayed step of bench.py --config $1 --batch $2 [library $3] as a timeline (gpurun_tools/timeline.py, anchored at clip+Adam)
cfg=${1:-native}; B=${2:-4096}; lib=${3:-gesture2vec_amd/libg2v_hip.so}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_tl
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 gpurun_tools/bench_altlib.py $lib --config $cfg --batch $B --steps 10 --warmup 5 --no-cpu-baseline --no-part-d > gpurun_out/prof_tl.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_tl/*/*kernel_trace.csv | head -1)
python gpurun_tools/timeline.py $f clip_adam | sed -n '/^ *0.0 dur/,$p' | cut -c1-120 > gpurun_out/r04_timeline_${cfg}_B${B}_$(basename $lib .so).txt
rm -rf gpurun_out/prof_tl
grep -v "dec_step\|pack_kernel\|fillBuffer" gpurun_out/r04_timeline_${cfg}_B${B}_$(basename $lib .so).txt | tail -48
