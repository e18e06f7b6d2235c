#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_t_pytest.log 2>&1
tail -3 gpurun_out/r06_t_pytest.log
bash gpurun_tools/r06_t2e_ab.sh
cp gpurun_out/r06_g_t2e_ab.log gpurun_out/r06_t_t2e_ab.log
for r in 1 2; do
  for args in "--config native --batch 4096 --steps 50" "--batch 8192 --config native --steps 30"; do
    for tree in ab_old .; do
      (cd $tree && timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$tree', '$args', d['ms_per_step'])")
    done
  done
done | tee gpurun_out/r06_t_engine_ab.log
