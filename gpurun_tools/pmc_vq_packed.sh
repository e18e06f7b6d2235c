#!/bin/bash
# HBM traffic of the fused VQ kernel (fragment-image form) at N=4096: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel trace only
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_vqp_$c -- python3 gpurun_tools/vq_only.py 4096 packed > gpurun_out/pmc_vqp_$c.log 2>&1
done
python3 - <<'P'
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_vqp_{c}/*/*counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "vq_fused_assign_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c] = (sum(vals) / len(vals), len(vals))
fetch_kb, n = out["FETCH_SIZE"]; write_kb, _ = out["WRITE_SIZE"]
N, E, K = 4096, 128, 512
res = {"N=4096": {"kernel": "void g2v::vq_fused_assign_kernel<128, true>", "dispatches": n,
                  "FETCH_SIZE_KB_per_launch_raw": round(fetch_kb, 2), "WRITE_SIZE_KB_per_launch_raw": round(write_kb, 2),
                  "hbm_bytes_per_launch_corrected": int((2 * fetch_kb + write_kb) * 1024),
                  "algorithmic_bytes_per_launch": N * (12 * E + 8) + 4 * E * E + 4 * E + 4 * K * E + 4 * K,
                  "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE exact. Separate rocprofv3 --pmc <C> --kernel-trace passes over gpurun_tools/vq_only.py 4096 packed (20 launches, all averaged)."}}
json.dump(res, open("gpurun_out/r02_vqp_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res))
P
