#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_text2embedding.py tests/test_gpu_loss_chase.py -x -q 2>&1 | tail -12
timeout 300 python gpurun_tools/bench_t2e.py > gpurun_out/r05_e_part_d_bench.json 2> gpurun_out/r05_e_part_d_bench.err; echo rc=$?
tail -c 1500 gpurun_out/r05_e_part_d_bench.err; tail -c 1200 gpurun_out/r05_e_part_d_bench.json
