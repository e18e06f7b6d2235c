#!/bin/bash
# Round 6: timing-experiment builds of the library that differ only in gru.hip's -DG2V_RES_DIAG=<n> (the W_hh-resident GRU forward):
#   gpurun_tools/r06_build_res_variants.sh 0 1 2 3 4   ->  gpurun_tools/libg2v_res<n>.so   (product objects for the rest)
set -e
cd "$(dirname "$0")/../gesture2vec_amd/csrc"
mkdir -p /tmp/resv
OTHERS=$(ls *.o | grep -v '^gru\.o$')
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DG2V_RES_DIAG=$v ${EXTRA} -c gru.hip -o /tmp/resv/gru_$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/resv/gru_$v.o $OTHERS -o ../../gpurun_tools/libg2v_res$v.so
  echo "built libg2v_res$v.so"
done
