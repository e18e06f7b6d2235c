#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
timeout 600 python -m pytest tests/test_gpu_loss_fold.py tests/test_gpu_ops.py -x -q -m gpu -k "quantiser_backward or vq_bwd or gru" 2>&1 < /dev/null | tail -3
for o in 1 0 1 0; do
  G2V_FUSE_VQ_BWD=$o timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'fuse_vq_bwd': $o, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))"
done
