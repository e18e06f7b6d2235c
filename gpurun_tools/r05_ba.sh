#!/bin/bash
mkdir -p gpurun_out
bash gpurun_tools/r04_tl_cfg.sh native 4096 > /dev/null 2>&1; cp gpurun_out/r04_timeline_native_B4096_libg2v_hip.txt gpurun_out/r05_ba_timeline_native_B4096.txt
grep -v "dec_step\|pack_kernel\|fillBuffer" gpurun_out/r05_ba_timeline_native_B4096.txt | tail -40 | cut -c1-140
