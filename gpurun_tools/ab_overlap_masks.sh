#!/bin/bash
# parallel-branch variants of the fused step: tests, then the bench line per G2V_OVERLAP mask
python -m pytest tests/test_gpu_vqvae.py -m gpu -q --tb=short -x -k "not 4096" 2>&1 | grep -v "where\|amdgpu" | tail -3
for m in 7 15 7 15; do
  echo "== G2V_OVERLAP=$m"
  G2V_OVERLAP=$m python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
