#!/bin/bash
mkdir -p gpurun_out
for k in 1 2; do timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "gru or dec_cluster or dec_rollout_fwd_bwd or dec_rollout_eval or teacher" 2>&1 | tail -4; done
timeout 600 python -m pytest tests/test_gpu_text2embedding.py tests/test_gpu_vqvae.py -x -q 2>&1 | tail -3
timeout 120 python gpurun_tools/stamps_dcl.py 2>/dev/null | tail -9
bash gpurun_tools/native_prof.sh 128 2>/dev/null | head -12
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | cut -c1-420 | tee gpurun_out/r05_q_part_d_bench.json
