#!/bin/bash
# diagnostic build: the product sources with -DG2V_PSTAMPS -> gpurun_tools/libg2v_pstamps.so (objects under /tmp)
set -e
cd "$(dirname "$0")/../gesture2vec_amd/csrc"
mkdir -p /tmp/pst
for f in linear vq gru dec_rollout dec_persist seq2seq misc; do
  if [ $f = dec_persist ] || [ ! -f /tmp/pst/$f.o ] || [ $f.hip -nt /tmp/pst/$f.o ] || [ common.hpp -nt /tmp/pst/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DG2V_PSTAMPS -c $f.hip -o /tmp/pst/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/pst/*.o -o ../../gpurun_tools/libg2v_pstamps.so
