#!/bin/bash
# A/B of the decoder weight-gradient launch order inside the default bench step
cd "${GRAFT_REPO_ROOT:?}"
for o in 0 1 0 1; do
  G2V_WGRAD_ORDER=$o timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'small_first': $o, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))"
done
