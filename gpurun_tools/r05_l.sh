#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_text2embedding.py tests/test_gpu_ops.py tests/test_gpu_train_script.py tests/test_gpu_data_path.py -x -q 2>&1 | tail -5
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | tee gpurun_out/r05_l_part_d_bench.json
