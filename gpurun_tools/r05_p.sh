#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "gru" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_text2embedding.py -x -q 2>&1 | tail -5
bash gpurun_tools/r04_prof_t2e.sh 128 False | head -12; mv gpurun_out/r04_e_kernel_stats_part_d_B128_attFalse.csv gpurun_out/r05_p_kernel_stats_part_d_B128_noatt.csv
timeout 300 python bench.py --config native --steps 200 --warmup 10 --no-cpu-baseline --no-part-d --sustained 0 2>/dev/null | tail -1 | cut -c1-300 | tee gpurun_out/r05_p_native.json
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | tee gpurun_out/r05_p_part_d_bench.json
