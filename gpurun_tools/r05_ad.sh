#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "linear or wgrad or weight" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_loss_chase.py -q -x 2>&1 | tail -4
for v in 1 0 1 0; do export G2V_FOLD_IN_GRAD=$v; echo "G2V_FOLD_IN_GRAD=$v"; timeout 300 python gpurun_tools/bench_native.py 2>/dev/null | tail -1; done
export G2V_FOLD_IN_GRAD=1
bash gpurun_tools/r04_tl_cfg.sh native 128 > gpurun_out/r05_ad_tl.log 2>&1; sed -n '/dec_cluster_bwd/,$p' gpurun_out/r05_ad_tl.log | cut -c1-140
