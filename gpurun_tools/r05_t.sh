#!/bin/bash
for k in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "dec_cluster or dec_rollout_fwd_bwd or dec_rollout_eval or teacher" 2>&1 | tail -6; done
