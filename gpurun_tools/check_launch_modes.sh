#!/bin/bash
# final state: default bench line (+ stderr), the DP path with one rank, torchrun single-rank launch as the driver does it
python bench.py > gpurun_out/r02_c_bench_default.json 2> gpurun_out/r02_c_bench_default.err; tail -c 400 gpurun_out/r02_c_bench_default.json; echo
python bench.py --force-dp --steps 60 --warmup 5 --no-cpu-baseline 2>gpurun_out/forcedp.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-dp', d['ms_per_step'], d['config']['launch'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline 2>gpurun_out/torchrun.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('torchrun n=1', d['ms_per_step'], d['n_gpus'], d['config']['launch'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
