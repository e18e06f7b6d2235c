#!/bin/bash
# encoder GRU weight gradients inside its backward kernel: parity, then A/B of the three modes
for m in 1 2; do
  echo "== tests, G2V_ENC_FUSED_WGRAD=$m"
  G2V_ENC_FUSED_WGRAD=$m timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_dp_engine.py -m gpu -q --tb=short -x 2>&1 | grep -v "where\|amdgpu" | tail -3
done
for rep in 1 2; do
  for m in 0 1 2; do
    G2V_ENC_FUSED_WGRAD=$m timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('enc fused mode $m:', d['ms_per_step'], d['config']['final_loss'])"
  done
done
