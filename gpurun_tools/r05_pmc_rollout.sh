#!/bin/bash
# round 5: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes; FETCH doubled per MI355X_MICROARCH.md) of the two persistent
# rollout kernels, the loss chaser and the quantiser kernel inside bench.py's step (eager launches: one dispatch record each).
# usage: r05_pmc_rollout.sh <tag>   -> gpurun_out/r05_<tag>_pmc_rollout.json
tag=${1:-a}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
i=0
for g in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf gpurun_out/pmcr_$i
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcr_$i -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-part-d --no-graph > gpurun_out/pmcr_$i.log 2>&1
  i=$((i+1))
done
python3 - "$tag" <<'P'
import csv, glob, json, collections, sys
tag = sys.argv[1]
keys = ("dec_persist_fwd_kernel", "dec_persist_bwd_kernel", "loss_chase_kernel", "vq_fused_bx_kernel", "gru_fwd_fast_kernel",
        "gru_bwd_fast_kernel", "gemm_tn_wave_kernel", "slab_reduce2_kernel", "gemm_nt_k4_kernel", "vq_stats_owner_kernel")
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmcr_*")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for key in keys:
                if key in n and "exact" not in n:
                    acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            res[k][c + "_KB_raw_avg"] = round(sum(v) / len(v), 1); res[k]["dispatches"] = len(v)
for k, v in res.items():
    v["hbm_MB_per_launch_corrected"] = round((2 * v.get("FETCH_SIZE_KB_raw_avg", 0) + v.get("WRITE_SIZE_KB_raw_avg", 0)) / 1024, 2)
    v["read_MB"] = round(2 * v.get("FETCH_SIZE_KB_raw_avg", 0) / 1024, 2); v["write_MB"] = round(v.get("WRITE_SIZE_KB_raw_avg", 0) / 1024, 2)
out = {"command": "rocprofv3 --pmc <C> --kernel-trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-part-d --no-graph",
       "note": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE exact; separate passes; B=4096,T=34,D=135,H=64",
       "kernels": res}
json.dump(out, open(f"gpurun_out/r05_{tag}_pmc_rollout.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
rm -rf gpurun_out/pmcr_*
