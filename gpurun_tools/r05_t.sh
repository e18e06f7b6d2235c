#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "dec_cluster" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "dec_rollout_fwd_bwd or teacher" 2>&1 | tail -8
