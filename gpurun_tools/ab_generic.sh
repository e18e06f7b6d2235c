#!/bin/bash
# quick check of a change: the engine parity tests at small sizes + three 200-step bench lines
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_dp_engine.py -m gpu -q --tb=short -k "not 4096" 2>&1 | grep -v "where\|amdgpu" | tail -3
for rep in 1 2 3; do timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"; done
