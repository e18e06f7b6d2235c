#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "linear_fwd_pair" 2>&1 | tail -8
bash gpurun_tools/r05_pmc_smallm_wgrad.sh
