#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "linear or wgrad or weight or gru or dec_cluster" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_text2embedding.py tests/test_gpu_vqvae.py -q 2>&1 | tail -3
for v in 1 0; do export G2V_SMW_WIDE=$v; echo "G2V_SMW_WIDE=$v"; bash gpurun_tools/r04_tl_cfg.sh native 128 > gpurun_out/r05_ac_tl_$v.log 2>&1; grep -E "smallm|period" gpurun_out/r05_ac_tl_$v.log
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | cut -c1-330; done
