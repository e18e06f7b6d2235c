#!/bin/bash
# kernel statistics of the fused soft-quantiser iteration (G2V_ONLY=1) at B = ${1:-4096}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
export G2V_ONLY=1
rm -rf gpurun_out/prof_gs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gs -- python3 gpurun_tools/gssoft_bench.py ${1:-4096} > gpurun_out/prof_gs.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_gs/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = max(int(r["Calls"]) for r in rows if "dec_persist_fwd" in r["Name"])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("iterations", n, "kernel time per iteration us", round(tot / n / 1e3, 1))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:64]:64s} calls/it {int(r["Calls"]) / n:5.1f} us/it {float(r["TotalDurationNs"]) / n / 1e3:8.1f}')
P
fi
rm -rf gpurun_out/prof_gs
