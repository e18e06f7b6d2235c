#!/bin/bash
mkdir -p gpurun_out/r2k
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -k "test_dec_rollout_fwd_bwd and 4096" 2>&1 | grep -E "passed|failed|Error" | tail -3 >> gpurun_out/r2k/loop.txt
done
echo "--- full ops file ---" >> gpurun_out/r2k/loop.txt
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line 2>&1 | grep -E "passed|failed|Error" | tail -4 >> gpurun_out/r2k/loop.txt
done
echo "--- no persist ---" >> gpurun_out/r2k/loop.txt
for i in 1 2; do
  G2V_NO_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=line -k "not persistent_matches" 2>&1 | grep -E "passed|failed|Error" | tail -4 >> gpurun_out/r2k/loop.txt
done
cat gpurun_out/r2k/loop.txt
