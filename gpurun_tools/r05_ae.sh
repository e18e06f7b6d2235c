#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "fold2 or row_mapped or small_row" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_vqvae.py -q -x -k "native or genea or generic" 2>&1 | tail -3
bash gpurun_tools/r04_tl_cfg.sh native 128 > gpurun_out/r05_ae_tl.log 2>&1; sed -n '/gru_cluster_bwd/,$p' gpurun_out/r05_ae_tl.log | cut -c1-140
timeout 300 python gpurun_tools/bench_native.py 2>/dev/null | tail -1
