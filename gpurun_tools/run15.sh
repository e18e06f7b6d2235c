#!/bin/bash
python - <<'PY'
import torch
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
for p in (-2, -1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p); print(p, "ok", s.priority)
    except Exception as e:
        print(p, "err", e)
PY
for pr in 0 1 2 -1; do
  echo "== side priority $pr"
  G2V_SIDE_PRIORITY=$pr python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2c -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r2c.log 2>&1
f=$(ls gpurun_out/prof_r2c/*/*kernel_trace.csv | head -1); python gpurun_tools/timeline.py $f > gpurun_out/timeline_r2c.txt; tail -45 gpurun_out/timeline_r2c.txt
