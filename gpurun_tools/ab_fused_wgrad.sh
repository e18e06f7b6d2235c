#!/bin/bash
# W_hh1 weight gradient fused into the persistent rollout backward: parity tests, then A/B against G2V_NO_FUSED_WGRAD=1
timeout 1200 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_dp_engine.py tests/test_gpu_ops.py -m gpu -q --tb=short 2>&1 | grep -v "where\|amdgpu" | tail -4
for rep in 1 2 3; do
  G2V_NO_FUSED_WGRAD=1 timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('separate:', d['ms_per_step'], d['config']['final_loss'])"
  timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused   :', d['ms_per_step'], d['config']['final_loss'])"
done
