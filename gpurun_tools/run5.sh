#!/bin/bash
mkdir -p gpurun_out/r2e
P=./gpurun_tools/px_test
for args in "256 33 0 0" "256 33 0 1" "256 33 168 1" "2 33 0 0" "20 33 50 1" "256 200 100 1"; do timeout 60 $P $args >> gpurun_out/r2e/px.txt 2>&1; done
timeout 120 python gpurun_tools/persist_diag.py 32 6 > gpurun_out/r2e/diag32.txt 2>&1
timeout 120 python gpurun_tools/persist_diag.py 4096 6 > gpurun_out/r2e/diag4096.txt 2>&1
cat gpurun_out/r2e/px.txt gpurun_out/r2e/diag32.txt gpurun_out/r2e/diag4096.txt
