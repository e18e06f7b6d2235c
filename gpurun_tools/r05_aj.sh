#!/bin/bash
mkdir -p gpurun_out
for M in 2560 2432 1280; do for c in 8 4; do export G2V_SMW_NW=$c; echo -n "NW=$c "; timeout 120 python gpurun_tools/wgrad_batch_bench.py $M 600 200 2>&1 | tail -1 | cut -c1-150; done; done
export G2V_SMW_NW=8
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "linear or wgrad or weight" 2>&1 | tail -3
for c in 8 4 8 4; do export G2V_SMW_NW=$c; echo "NW=$c"; timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 --config native --steps 300 2>/dev/null | tail -1 | cut -c1-140; done
export G2V_SMW_NW=8
bash gpurun_tools/r04_tl_cfg.sh native 128 > /dev/null 2>&1; sed -n '/dec_cluster_bwd/,$p' gpurun_out/r04_timeline_native_B128_libg2v_hip.txt | cut -c1-130
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | cut -c1-200
