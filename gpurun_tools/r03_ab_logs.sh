#!/bin/bash
# round-3 A/B evidence kept under profiles/: the custom_loss fold (phase stamps + bench line), the co-run experiment, the launch orders
cd "${GRAFT_REPO_ROOT:?}"
{
  echo "# gpurun_tools/fold_ab.sh: custom_loss carried by the rollout pair (G2V_LOSS_FOLD=1) vs its own launch (0)"
  bash gpurun_tools/fold_ab.sh 2>&1
} > gpurun_out/r03_loss_fold_ab.log
{
  echo "# gpurun_tools/corun_probe.py: a side-stream kernel beside the encoder BPTT (eager launches, events)"
  timeout 300 python gpurun_tools/corun_probe.py 2>&1 < /dev/null | tail -2
  echo "# G2V_WGRAD_ORDER (1 = the small product first, round 2's order) inside the default bench step"
  bash gpurun_tools/order_ab.sh 2>&1
  echo "# G2V_FORK_ORDER (bit k: branch k launched behind the main chain's next kernel) inside the default bench step"
  bash gpurun_tools/fork_ab.sh 2>&1
} > gpurun_out/r03_corun_and_launch_order_ab.log
tail -5 gpurun_out/r03_loss_fold_ab.log; tail -9 gpurun_out/r03_corun_and_launch_order_ab.log
