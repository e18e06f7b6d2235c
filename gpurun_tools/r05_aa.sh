#!/bin/bash
mkdir -p gpurun_out
for M in 2560 2432 640; do for c in 0 22 24 42 33; do export G2V_SMW_LDS=$c; echo -n "LDS=$c "; timeout 120 python gpurun_tools/wgrad_batch_bench.py $M 600 200 2>&1 | tail -1; done; done
export G2V_SMW_LDS=22
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "linear or wgrad or weight" 2>&1 | tail -3
for v in 1 0; do export G2V_GRU_CL_BWD_L2X=$v; echo "G2V_GRU_CL_BWD_L2X=$v"; bash gpurun_tools/r04_tl_cfg.sh native 128 > gpurun_out/r05_aa_tl_$v.log 2>&1; grep -E "gru_cluster|smallm|period" gpurun_out/r05_aa_tl_$v.log; done
