#!/bin/bash
# bench + stamps + vq tests of the bf16-screened fused VQ kernel (round 3 iteration loop)
mkdir -p gpurun_out
python gpurun_tools/vq_bx_bench.py ${1:-4096} > gpurun_out/vq_bx_bench.log 2>&1
python - <<PY
import json
for r in json.load(open("gpurun_out/vq_bx_bench.json")):
    print(r["N"], r["data"], {k: v for k, v in r.items() if k.startswith("us_") or k.startswith("frac")}, r["bx2"]["idx_mismatch"], r["bx2"]["flat_idx_quant_sse_bitwise"], r["bx2"]["exact_tiles"], r["bx2"]["pairs"])
PY
python gpurun_tools/vqstamps_bx.py 0 > gpurun_out/vqstamps_bx.log 2>&1; grep -v amdgpu.ids gpurun_out/vqstamps_bx.log
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "vq" 2>&1 | tail -5
