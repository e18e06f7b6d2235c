#!/bin/bash
# SQ counters of the bulk-assignment sweep kernel, one counter group per pass
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_bulk_$i -- python3 gpurun_tools/bulk_only.py > gpurun_out/pmc_bulk_$i.log 2>&1
  f=$(ls gpurun_out/pmc_bulk_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    if "bx3_sweep" in k or "bulk_sweep" in k or "vq_assign_rt" in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    for c, v in d.items():
        print(k, c, v / max(n[(k, c)], 1))
PY
done
