#!/bin/bash
mkdir -p gpurun_out
for np in 2 4; do for M in 2560 640; do for c in 22 21 11; do export G2V_SMW_TILE=$c; echo -n "TILE=$c "; timeout 120 python gpurun_tools/wgrad_batch_bench.py $M 600 200 $np 2>&1 | tail -1 | cut -c1-150; done; done; done
export G2V_SMW_TILE=0
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "fold2 or small_row" 2>&1 | tail -3
for c in 0 22 0 22; do export G2V_SMW_TILE=$c; echo "TILE=$c"; timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 --config native --steps 300 2>/dev/null | tail -1 | cut -c1-140; done
export G2V_SMW_TILE=0
bash gpurun_tools/r04_tl_cfg.sh native 128 > /dev/null 2>&1; sed -n '/gru_cluster_bwd/,$p' gpurun_out/r04_timeline_native_B128_libg2v_hip.txt | cut -c1-130
