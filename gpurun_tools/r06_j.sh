#!/bin/bash
# attention kernels with their loads up front: operator tests + Part d tests, then same-box A/B against ab_old/
cd "${GRAFT_REPO_ROOT:?}"
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_text2embedding.py -x -q -m gpu  > gpurun_out/r06_k_pytest.log 2>&1
tail -3 gpurun_out/r06_k_pytest.log
bash gpurun_tools/r06_t2e_ab.sh
cp gpurun_out/r06_g_t2e_ab.log gpurun_out/r06_k_t2e_ab.log
