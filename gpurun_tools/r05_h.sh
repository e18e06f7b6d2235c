#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "vq_assign" 2>&1 | tail -5
timeout 300 python gpurun_tools/r05_vq400_bench.py 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_vqvae.py -q --tb=line -rP -k "fused_train_step_vs_oracle or report_large" 2>&1 | grep -E "Error|assert|worst|passed|failed" | cut -c1-260 | tail -40
