#!/bin/bash
mkdir -p gpurun_out
for M in 2560 640; do for c in 22 122 112 12 121; do export G2V_SMW_LDS=$c; echo -n "LDS=$c "; timeout 120 python gpurun_tools/wgrad_batch_bench.py $M 600 200 2>&1 | tail -1 | cut -c1-120; done; done
