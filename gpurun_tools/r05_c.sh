#!/bin/bash
# round 5, step c: first GPU run of the fused Part-d decoder-step kernels + the saved-tensor diet probe + Part d timelines
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_text2embedding.py -x -q 2>&1 | tail -15
timeout 300 python gpurun_tools/bench_t2e.py 2>&1 | tail -2 | tee gpurun_out/r05_c_part_d_bench.json
timeout 300 python gpurun_tools/r05_saved_diet_probe.py 2>&1 | tail -2 | tee gpurun_out/r05_c_saved_diet_probe.json
bash gpurun_tools/t2e_tl.sh 128 False > /dev/null 2>&1; cp gpurun_out/t2e_timeline_B128_attFalse.txt gpurun_out/r05_c_t2e_timeline_B128_noatt.txt
bash gpurun_tools/t2e_tl.sh 4096 False > /dev/null 2>&1; cp gpurun_out/t2e_timeline_B4096_attFalse.txt gpurun_out/r05_c_t2e_timeline_B4096_noatt.txt
wc -l gpurun_out/r05_c_t2e_timeline_*.txt
