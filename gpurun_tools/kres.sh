#!/bin/bash
# kernel resource usage of one kernel of a csrc file:  gpurun_tools/kres.sh vq.hip vq_fused_bx_kernelP [-DFOO=1 ...]
f=$1; k=$2; shift 2
cd "$(dirname "$0")/../gesture2vec_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 \
  | grep -A12 "Function Name: _Z[0-9]*$k" | grep -E "Function Name|VGPRs:|AGPRs|Spill|LDS Size|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '; echo
