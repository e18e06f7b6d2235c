#!/bin/bash
# round 6, Part d launch diet: tests of the touched paths, Part d bench (eager + graph), timelines at B = 128
cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_flat_optimizer.py tests/test_gpu_text2embedding.py -x -q -m gpu > gpurun_out/r06_g_pytest.log 2>&1
tail -5 gpurun_out/r06_g_pytest.log
timeout 600 python gpurun_tools/bench_t2e.py > gpurun_out/r06_g_bench_t2e.json 2> gpurun_out/r06_g_bench_t2e.err
cat gpurun_out/r06_g_bench_t2e.json
bash gpurun_tools/t2e_tl.sh 128 False > /dev/null 2>&1
bash gpurun_tools/t2e_tl.sh 128 True > /dev/null 2>&1
cp gpurun_out/t2e_timeline_B128_attFalse.txt gpurun_out/r06_g_t2e_timeline_B128_attFalse.txt
cp gpurun_out/t2e_timeline_B128_attTrue.txt gpurun_out/r06_g_t2e_timeline_B128_attTrue.txt
tail -3 gpurun_out/r06_g_t2e_timeline_B128_attFalse.txt
