#!/bin/bash
mkdir -p gpurun_out
for k in 1 2; do timeout 600 python bench.py --no-part-d 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['steps'], d['warmup'], d['ms_per_step'], d['value'], d['sustained'], d['cpu_baseline']['value'])"; done
timeout 900 python bench.py > gpurun_out/r05_bb_bench_default.json 2>/dev/null; tail -1 gpurun_out/r05_bb_bench_default.json | cut -c1-200
