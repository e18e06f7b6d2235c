#!/bin/bash
# loss fold: parity tests, phase stamps and A/B of the default bench line (fold on / off)
cd "${GRAFT_REPO_ROOT:?}"
timeout 600 python -m pytest tests/test_gpu_loss_fold.py -x -q -m gpu 2>&1 | tail -5
for f in 1 0; do echo "== fold $f"; G2V_LOSS_FOLD=$f timeout 250 python gpurun_tools/pstamps.py 2>&1 | grep -A1 "slot 0\|slot 2" | grep -v "^--"; done
for f in 1 0 1 0; do
  G2V_LOSS_FOLD=$f timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'fold': $f, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))"
done
