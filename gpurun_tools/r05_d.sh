#!/bin/bash
# round 5, step d: full GPU suite after the fused Part-d kernels + deferred commits; Part d bench; saved-tensor diet probe; timelines
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r05_d_gpu_tests.log
timeout 300 python gpurun_tools/bench_t2e.py 2>&1 | tail -1 | tee gpurun_out/r05_d_part_d_bench.json
timeout 300 python gpurun_tools/r05_saved_diet_probe.py 2>&1 | tail -2 | tee gpurun_out/r05_d_saved_diet_probe.json
bash gpurun_tools/t2e_tl.sh 4096 False > /dev/null 2>&1; cp gpurun_out/t2e_timeline_B4096_attFalse.txt gpurun_out/r05_d_t2e_timeline_B4096_noatt.txt
bash gpurun_tools/t2e_tl.sh 4096 True > /dev/null 2>&1; cp gpurun_out/t2e_timeline_B4096_attTrue.txt gpurun_out/r05_d_t2e_timeline_B4096_att.txt
timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-part-d 2>/dev/null | tail -1 | cut -c1-300
