#!/bin/bash
# SQ / LDS / TCP counters of the generic-shape kernels (bench.py --config native --batch 4096): per-dispatch averages per kernel
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum"
)
i=0
for g in "${groups[@]}"; do
  rm -rf gpurun_out/pmcn_$i
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcn_$i -- python3 bench.py --config native --batch 4096 --steps 3 --warmup 2 --no-cpu-baseline --no-part-d --no-graph > gpurun_out/pmcn_$i.log 2>&1
  i=$((i+1))
done
python3 - <<'P'
import csv, glob, json, collections
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmcn_*")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for key in ("gru_seq_bwd_kernel", "gru_seq_fwd_kernel", "dec_step_fwd_kernel", "dec_step_bwd_kernel", "gemm_tn_wave_gen_kernel<4"):
                if key in n:
                    acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            res[k][c] = round(sum(v) / len(v), 1)
json.dump(res, open("gpurun_out/r04_pmc_native.json", "w"), indent=1)
print(json.dumps(res, indent=1))
P
rm -rf gpurun_out/pmcn_*
