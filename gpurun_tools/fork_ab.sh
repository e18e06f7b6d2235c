#!/bin/bash
# A/B of the fork order per branch (bit k of G2V_FORK_ORDER: branch k launched behind the main chain's next kernel)
cd "${GRAFT_REPO_ROOT:?}"
for o in 0 1 2 3 0 1 2 3; do
  G2V_FORK_ORDER=$o timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'fork_late_mask': $o, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))"
done
