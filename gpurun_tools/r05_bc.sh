#!/bin/bash
mkdir -p gpurun_out
bash gpurun_tools/r05_pmc_smallm_wgrad.sh | tail -40
