#!/bin/bash
# PMC collection for the VQ kernels of the product path (separate passes, as MI355X_MICROARCH.md prescribes) + N sweep
mkdir -p gpurun_out/r2l
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_INSTS_VALU \b" | head -40 > $GRAFT_REPO_ROOT/gpurun_out/r2l/counters.txt
for which in fused stats; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40)
    rm -rf /tmp/pmc_$which_$tag
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${which}_$tag -o p -- python3 $GRAFT_REPO_ROOT/gpurun_tools/vq_only.py 4096 $which > /tmp/pmc.log 2>&1
    f=$(find /tmp/pmc_${which}_$tag -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then grep -E "Counter_Name|vq_fused|vq_stats_owner" $f | head -400 > $GRAFT_REPO_ROOT/gpurun_out/r2l/pmc_${which}_$tag.csv; else tail -3 /tmp/pmc.log > $GRAFT_REPO_ROOT/gpurun_out/r2l/pmc_${which}_$tag.err; fi
  done
done
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/r2l; head -3 gpurun_out/r2l/pmc_fused_FETCH_SIZE.csv; cat gpurun_out/r2l/counters.txt | head -30
