#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "preclear or dec_cluster or dec_rollout" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_loss_chase.py -q -x 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 --config native --steps 300 2>/dev/null | tail -1 | cut -c1-160
