#!/bin/bash
# Diagnostic counter passes (round 3): why does the fused VQ kernel's request stream run at ~25 B/clk/CU when a bare burst of the
# same size reaches 66?  One --pmc group per pass (kernel trace only), for the bx kernel and for the microbenchmark burst.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
groups=(
 "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAVES"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcd_bx_$i -- python3 gpurun_tools/vq_bx_only.py 0 > gpurun_out/pmcd_bx_$i.log 2>&1
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcd_mb_$i -- ./gpurun_tools/l2_stream_bench2 > gpurun_out/pmcd_mb_$i.log 2>&1
  i=$((i+1))
done
python3 - <<'P'
import csv, glob, json, collections
out = {}
for tag, pat in (("bx", "vq_fused_bx_kernel"), ("burst8x24", "burst_kernel<8, 24>")):
    res = collections.OrderedDict()
    for d in sorted(glob.glob(f"gpurun_out/pmcd_{'bx' if tag == 'bx' else 'mb'}_*")):
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if pat in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                res[k] = round(sum(v) / len(v), 1)
    out[tag] = res
json.dump(out, open("gpurun_out/pmc_diag.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
