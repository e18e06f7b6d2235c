#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "vq_assign or bulk" 2>&1 | tail -6
timeout 600 python gpurun_tools/bulk_assign_bench.py 2>&1 | tail -14
timeout 900 python -m pytest tests/test_gpu_text2embedding.py tests/test_gpu_thin_models.py tests/test_gpu_data_path.py -x -q -rP 2>&1 | grep -E "worst gradient|passed|failed|Error" | cut -c1-250 | tail -8
