#!/bin/bash
# round 6: deferred slab reductions -- the new tests, then same-box A/B of the default bench line (3 x 300 steps, alternating):
# immediate reductions | deferred | deferred + the encoder's two products on two branches (G2V_OVERLAP bit 4 at H = 64)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vqvae.py tests/test_gpu_text2embedding.py -q -m gpu -x -k "deferred or graphed_step or fault or h200 or shipped_width" 2>&1 | tail -4
out=gpurun_out/r06_c_defer_ab.log; : > $out
for rep in 1 2 3; do
  for cfg in "G2V_DEFER_REDUCE=0" "G2V_DEFER_REDUCE=1" "G2V_DEFER_REDUCE=1 G2V_OVERLAP=31"; do
    echo -n "$cfg  " >> $out
    env $cfg timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-part-d --sustained 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])" >> $out
  done
done
cat $out
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r6c -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-part-d --sustained 0 > gpurun_out/prof_r6c.log 2>&1
f=$(ls gpurun_out/prof_r6c/*/*kernel_trace.csv | head -1); python gpurun_tools/timeline.py $f > gpurun_out/r06_c_step_timeline.txt; tail -42 gpurun_out/r06_c_step_timeline.txt
rm -rf gpurun_out/prof_r6c
