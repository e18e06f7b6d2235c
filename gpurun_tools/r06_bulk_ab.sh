#!/bin/bash
# round 6: per-kernel time of g2v_vq_assign_bulk at 2^20 rows for several builds of the library (gpurun_tools/libg2v_bx<v>.so)
for v in "$@"; do bash gpurun_tools/r05_bulk_prof.sh 20 gpurun_tools/libg2v_bx$v.so 2>&1 | grep -i "sweep"; done
