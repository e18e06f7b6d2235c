#!/bin/bash
mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r2b/pytest.txt
E=./gpurun_tools/exchange_bench
for args in "0 0 0 33 0 16" "3 0 0 33 0 16" "6 0 0 33 0 16" "5 0 0 33 0 8" "5 0 0 33 0 16" "5 0 0 33 0 32" "4 0 0 33 0 8" "4 0 0 33 0 16" "4 0 0 33 0 32" \
            "0 168 64 33 0 16" "3 168 64 33 0 16" "6 168 64 33 0 16" "4 168 64 33 0 8" "4 168 64 33 0 16" "4 168 64 33 0 32" "4 168 64 33 1 16" "4 168 64 33 1 32" "6 168 64 33 1 16"; do
  timeout 60 $E $args >> gpurun_out/r2b/exchange.txt 2>&1
done
cat gpurun_out/r2b/pytest.txt gpurun_out/r2b/exchange.txt
