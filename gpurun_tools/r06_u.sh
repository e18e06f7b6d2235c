#!/bin/bash
# the streaming BPTT with global (not flat) prefetch loads: engine configs A/B (ab_old = previous commit) + the GRU tests
cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gru" > gpurun_out/r06_u_pytest.log 2>&1
tail -2 gpurun_out/r06_u_pytest.log
for r in 1 2; do
  for args in "--config native --batch 4096 --steps 50" "--config genea --steps 50" "--config native --batch 2048 --steps 50"; do
    for tree in ab_old .; do
      (cd $tree && timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$tree', '$args', d['ms_per_step'])")
    done
  done
done | tee gpurun_out/r06_u_engine_ab.log
