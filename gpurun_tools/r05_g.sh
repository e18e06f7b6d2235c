#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_vqvae.py -q -rP -k "fused_train_step_vs_oracle or report_large" 2>&1 | grep -v "^$" | tail -40
timeout 900 python -m pytest tests/test_gpu_loss_chase.py tests/test_gpu_text2embedding.py tests/test_gpu_ops.py -x -q 2>&1 | tail -5
for rep in 1 2 3; do for v in 0 1; do
  echo -n "GRUF_PACK_SIDE=$v "; G2V_GRUF_PACK_SIDE=$v timeout 300 python gpurun_tools/bench_attr.py --steps 300 --warmup 10 --no-cpu-baseline --no-part-d --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
done; done | tee gpurun_out/r05_g_gruf_pack_ab.log
timeout 300 python gpurun_tools/bench_t2e.py 2>/dev/null | tail -1 | tee gpurun_out/r05_g_part_d_bench.json
