#!/bin/bash
# HBM traffic + time of vq_stats_owner_kernel at N=4096 (separate FETCH_SIZE / WRITE_SIZE passes over gpurun_tools/vq_only.py 4096 stats)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcs_$c -- python3 gpurun_tools/vq_only.py 4096 stats > gpurun_out/pmcs_$c.log 2>&1
done
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmcs_t -- python3 gpurun_tools/vq_only.py 4096 stats > gpurun_out/pmcs_t.log 2>&1
python3 - <<'P'
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmcs_{c}/*/*counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "vq_stats_owner_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c] = sum(vals) / len(vals)
f = glob.glob("gpurun_out/pmcs_t/*/*kernel_stats.csv")[0]
avg = [float(r["AverageNs"]) for r in csv.DictReader(open(f)) if "vq_stats_owner_kernel" in r["Name"]][0]
N, E, K = 4096, 128, 512
res = {"kernel": "vq_stats_owner_kernel", "N": N, "avg_us": round(avg / 1e3, 2), "FETCH_SIZE_KB_raw": round(out["FETCH_SIZE"], 1), "WRITE_SIZE_KB_raw": round(out["WRITE_SIZE"], 1),
       "hbm_bytes_per_launch_corrected": int((2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024),
       "algorithmic_bytes_per_launch": N * E * 4 + N * 8 + K * E * 4 + K * 4}
res["traffic_over_algorithmic"] = round(res["hbm_bytes_per_launch_corrected"] / res["algorithmic_bytes_per_launch"], 2)
json.dump(res, open("gpurun_out/r03_vq_stats_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res))
P
rm -rf gpurun_out/pmcs_*
