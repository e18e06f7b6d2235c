#!/bin/bash
# round-2 GPU session 1: full GPU test suite, exchange-protocol microbench, default bench line
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a/pytest.txt
for v in 0 1 2; do for sk in 0 1; do
  timeout 60 ./gpurun_tools/exchange_bench $v 168 64 33 $sk >> gpurun_out/r2a/exchange.txt 2>&1
done; done
timeout 60 ./gpurun_tools/exchange_bench 0 264 64 33 0 >> gpurun_out/r2a/exchange.txt 2>&1
timeout 60 ./gpurun_tools/exchange_bench 1 264 64 33 0 >> gpurun_out/r2a/exchange.txt 2>&1
timeout 60 ./gpurun_tools/exchange_bench 2 264 64 33 0 >> gpurun_out/r2a/exchange.txt 2>&1
timeout 60 ./gpurun_tools/exchange_bench 1 0 0 33 0 >> gpurun_out/r2a/exchange.txt 2>&1
timeout 60 ./gpurun_tools/exchange_bench 2 0 0 33 0 >> gpurun_out/r2a/exchange.txt 2>&1
timeout 60 ./gpurun_tools/exchange_bench 1 168 0 33 0 >> gpurun_out/r2a/exchange.txt 2>&1
timeout 60 ./gpurun_tools/exchange_bench 2 168 0 33 0 >> gpurun_out/r2a/exchange.txt 2>&1
timeout 600 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
timeout 300 python bench.py --dropout 0.2 --no-cpu-baseline > gpurun_out/r2a/bench_p02.json 2>> gpurun_out/r2a/bench.err
cat gpurun_out/r2a/pytest.txt gpurun_out/r2a/exchange.txt gpurun_out/r2a/bench.json
