#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
timeout 1200 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_dp_engine.py tests/test_gpu_loss_fold.py tests/test_gpu_decoder_step.py -x -q -m gpu 2>&1 < /dev/null | tail -3
for o in 1 0 1 0; do
  G2V_ENC_FUSED_IN=$o timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'enc_fused_in': $o, 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'loss': d['config']['final_loss']}))"
done
for o in 1 0; do
  G2V_ENC_FUSED_IN=$o timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --dropout 0.2 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'dropout enc_fused_in': $o, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))"
done
