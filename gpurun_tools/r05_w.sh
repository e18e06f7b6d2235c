#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_vqvae.py tests/test_gpu_loss_chase.py tests/test_gpu_dp_engine.py tests/test_gpu_train_script.py tests/test_gpu_shipped_gssoft.py tests/test_gpu_data_path.py -q -x 2>&1 | tail -5
for cfg in "--config native" "--config genea --batch 128" ""; do
  r=$(timeout 300 python bench.py $cfg --steps 300 --warmup 10 --no-cpu-baseline --no-part-d --sustained 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['graph_branches_mask'])")
  echo "$cfg : $r"
done
