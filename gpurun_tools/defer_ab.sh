#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_gpu_dp_engine.py tests/test_gpu_train_script.py tests/test_gpu_loss_fold.py -x -q -m gpu 2>&1 < /dev/null | tail -3
for o in 1 0 1 0; do
  G2V_DEFER_STATS=$o timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'defer_stats': $o, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))"
done
for o in 1 0; do
  G2V_DEFER_STATS=$o timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --force-dp 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'force_dp defer_stats': $o, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))"
done
