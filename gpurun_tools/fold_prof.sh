#!/bin/bash
# per-kernel durations of the bench step with the loss fold on / off
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for f in 1 0; do
  export G2V_LOSS_FOLD=$f
  rm -rf gpurun_out/prof_fold$f
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fold$f -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_fold$f.log 2>&1
  s=$(ls gpurun_out/prof_fold$f/*/*kernel_stats.csv | head -1)
  echo "== fold=$f"; grep -i "dec_persist\|custom_loss" $s | cut -c1-200
  k=$(ls gpurun_out/prof_fold$f/*/*kernel_trace.csv | head -1); python gpurun_tools/timeline.py $k > gpurun_out/fold${f}_timeline.txt; tail -1 gpurun_out/fold${f}_timeline.txt
  rm -rf gpurun_out/prof_fold$f
done
