#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "gru or linear or wgrad or weight" 2>&1 | tail -3
for v in 1 0 1 0; do export G2V_SMW_XCD=$v; echo "G2V_SMW_XCD=$v"; bash gpurun_tools/native_prof.sh 128 2>/dev/null | grep -E "native|gru_cluster|smallm"; done
export G2V_SMW_XCD=1
bash gpurun_tools/r04_tl_cfg.sh native 128 > gpurun_out/r05_z_tl.log 2>&1; tail -60 gpurun_out/r05_z_tl.log
