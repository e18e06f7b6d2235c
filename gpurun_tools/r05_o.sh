#!/bin/bash
# GRU cluster kernels (one persistent launch for all steps at small batch): parity, then the two small-batch configurations they serve
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gru" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_text2embedding.py tests/test_gpu_vqvae.py -x -q 2>&1 | tail -5
timeout 300 python bench.py --config native --steps 200 --warmup 10 --no-cpu-baseline --no-part-d --sustained 0 2>/dev/null | tail -1 | cut -c1-400 | tee gpurun_out/r05_o_native.json
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | tee gpurun_out/r05_o_part_d_bench.json
