#!/bin/bash
# SQ / LDS / TCP counters of the LDS-staged small-row-count weight-gradient kernel (four 600 x 200 products at 2560 rows,
# gpurun_tools/wgrad_batch_bench.py), per-dispatch averages, one --pmc group per pass
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum"
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INST_LEVEL_LDS"
)
i=0
for g in "${groups[@]}"; do
  rm -rf gpurun_out/pmcw_$i
  timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcw_$i -- python3 gpurun_tools/wgrad_batch_bench.py 2560 600 200 > gpurun_out/pmcw_$i.log 2>&1
  i=$((i+1))
done
python3 - <<'P'
import csv, glob, json, collections
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmcw_*")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_tn_smallm_lds_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in acc.items():
            res["gemm_tn_smallm_lds_kernel<2,1,4> 4 x (2560 x 600)^T (2560 x 200)"][c] = round(sum(v) / len(v), 1)
json.dump(res, open("gpurun_out/r05_pmc_smallm_wgrad.json", "w"), indent=1)
print(json.dumps(res, indent=1))
P
rm -rf gpurun_out/pmcw_*
