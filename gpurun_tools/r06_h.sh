#!/bin/bash
# round 6, side branches of Part d's backward: tests, then the same-box A/B against ab_old/ (the previous commit)
cd "${GRAFT_REPO_ROOT:?}"
timeout 1200 python -m pytest tests/test_gpu_text2embedding.py tests/test_gpu_flat_optimizer.py -x -q -m gpu > gpurun_out/r06_h_pytest.log 2>&1
tail -4 gpurun_out/r06_h_pytest.log
bash gpurun_tools/r06_t2e_ab.sh
cp gpurun_out/r06_g_t2e_ab.log gpurun_out/r06_h_t2e_ab.log
