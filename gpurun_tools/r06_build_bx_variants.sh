#!/bin/bash
# Round 6: A/B builds of the library that differ only in vq.hip's -D$MACRO=<n> (default MACRO=G2V_BX_VARIANT, commit c2b89a7's switches;
# MACRO=G2V_BULK_DIAG: timing experiments of the bulk sweep) (and, with "s" appended, -DG2V_VQSTAMPS):
#   gpurun_tools/r06_build_bx_variants.sh 0 1 2 3 0s 3s   ->  gpurun_tools/libg2v_bx<n>.so
# The other objects are the product build's (gesture2vec_amd/csrc/*.o); run `make -C gesture2vec_amd/csrc` first.
set -e
cd "$(dirname "$0")/../gesture2vec_amd/csrc"
mkdir -p /tmp/bxv
OTHERS=$(ls *.o | grep -v '^vq\.o$')
for v in "$@"; do
  n=${v%s}; extra=""
  [ "$v" != "$n" ] && extra="-DG2V_VQSTAMPS"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -D${MACRO:-G2V_BX_VARIANT}=$n $extra -c vq.hip -o /tmp/bxv/vq_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/bxv/vq_$v.o $OTHERS -o ../../gpurun_tools/libg2v_bx$v.so
  echo "built libg2v_bx$v.so"
done
