#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -X faulthandler gpurun_tools/bench_t2e.py > gpurun_out/r05_f.json 2> gpurun_out/r05_f.err; echo rc=$?
grep -v Warning gpurun_out/r05_f.err | tail -60
