#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the bf16-screened fused VQ kernel at N=4096
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS"; do
  timeout 150 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcx_$i -- python3 gpurun_tools/vq_bx_only.py 0 > gpurun_out/pmcx_$i.log 2>&1
  i=$((i+1))
done
python3 - <<'P'
import csv, glob, json, collections
res = collections.OrderedDict(); n = 0
for d in sorted(glob.glob("gpurun_out/pmcx_*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "vq_fused_bx_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k] = round(sum(v) / len(v), 2); n = len(v)
N, E, K = 4096, 128, 512
out = {"N=4096": {"kernel": "vq_fused_bx_kernel", "dispatches": n,
                  "FETCH_SIZE_KB_per_launch_raw": res.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch_raw": res.get("WRITE_SIZE"),
                  "hbm_bytes_per_launch_corrected": int((2 * res.get("FETCH_SIZE", 0) + res.get("WRITE_SIZE", 0)) * 1024),
                  "algorithmic_bytes_per_launch": N * (12 * E + 8) + 4 * E * E + 4 * E + 4 * K * E + 4 * K,
                  "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE exact. Separate rocprofv3 --pmc <C> --kernel-trace passes over gpurun_tools/vq_bx_only.py (20 launches, all averaged)."},
       "sq_counters_per_launch": {k: v for k, v in res.items() if k.startswith("SQ_")}}
json.dump(out, open("gpurun_out/r03_vqbx_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
rm -rf gpurun_out/pmcx_*
