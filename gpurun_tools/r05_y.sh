#!/bin/bash
mkdir -p gpurun_out
for k in 1 2; do timeout 900 python -m pytest tests/test_gpu_text2embedding.py -q 2>&1 | tail -3; done
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | cut -c1-700 | tee gpurun_out/r05_y_part_d_bench.json
bash gpurun_tools/r04_prof_t2e.sh 128 False | head -14; mv gpurun_out/r04_e_kernel_stats_part_d_B128_attFalse.csv gpurun_out/r05_y_kernel_stats_part_d_B128_noatt.csv
for v in 3 0 1 2 3 0; do export G2V_GRU_CL_L2X=$v; echo "G2V_GRU_CL_L2X=$v"; bash gpurun_tools/native_prof.sh 128 2>/dev/null | grep -E "native|gru_cluster|dec_cluster"; done
