#!/bin/bash
# W_hh-resident GRU BPTT: same-box A/B (ab_old = previous commit): Part d and the H = 200 engine configs
cd "${GRAFT_REPO_ROOT:?}"
bash gpurun_tools/r06_t2e_ab.sh
cp gpurun_out/r06_g_t2e_ab.log gpurun_out/r06_n_t2e_ab.log
for r in 1 2; do
  for args in "--config native --batch 4096 --steps 50" "--config genea --steps 50" "--config native --batch 2048 --steps 50"; do
    for tree in ab_old .; do
      (cd $tree && timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$tree', '$args', d['ms_per_step'])")
    done
  done
done | tee gpurun_out/r06_n_engine_ab.log
