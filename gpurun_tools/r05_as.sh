#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_text2embedding.py -q -x 2>&1 | tail -4
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | cut -c1-900 | tee gpurun_out/r05_as_part_d_bench.json
bash gpurun_tools/r04_prof_t2e.sh 4096 False | head -16; mv gpurun_out/r04_e_kernel_stats_part_d_B4096_attFalse.csv gpurun_out/r05_as_kernel_stats_part_d_B4096_noatt.csv
