#!/bin/bash
# Round 6: A/B of the G2V_BX_VARIANT builds (gpurun_tools/r06_build_bx_variants.sh) on one box: timing + bitwise checks per build,
# twice in alternating order, then the in-kernel stamps of the stamp builds.  usage: r06_bx_ab.sh "0 1 2 ..." "0s 7s ..."
mkdir -p gpurun_out
out=gpurun_out/r06_bx_ab.log; : > $out
for rep in 1 2; do
  for v in $1; do
    timeout 300 python gpurun_tools/r06_bx_variant_bench.py gpurun_tools/libg2v_bx$v.so 4096 2>&1 | grep '^{' >> $out
  done
done
for v in $2; do
  timeout 300 python gpurun_tools/vqstamps_bx.py 0 gpurun_tools/libg2v_bx$v.so 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
