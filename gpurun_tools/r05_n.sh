#!/bin/bash
# Part d after removing the torch-side copies (outputs cat, shifted-state cats, strided logits copy): tests, bench, kernel stats
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_text2embedding.py tests/test_gpu_thin_models.py tests/test_gpu_train_script.py -x -q 2>&1 | tail -5
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | tee gpurun_out/r05_n_part_d_bench.json
bash gpurun_tools/r04_prof_t2e.sh 4096 False | head -30; mv gpurun_out/r04_e_kernel_stats_part_d_B4096_attFalse.csv gpurun_out/r05_n_kernel_stats_part_d_B4096_noatt.csv
