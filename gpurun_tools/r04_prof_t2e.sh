#!/bin/bash
# kernel statistics of Part d (text2embedding train iterations, gpurun_tools/prof_t2e.py) at B = $1, attention $2
B=${1:-4096}; att=${2:-False}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_t2e
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t2e -- python3 gpurun_tools/prof_t2e.py $B $att > gpurun_out/prof_t2e.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_t2e/*/*kernel_stats.csv 2>/dev/null | head -1)
cp $f gpurun_out/r04_e_kernel_stats_part_d_B${B}_att${att}.csv
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 10 iterations", round(tot / 1e6, 1))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    print(f'{r["Name"][:84]:84s} calls {int(r["Calls"]):6d} total_ms {float(r["TotalDurationNs"]) / 1e6:8.2f} avg_us {float(r["AverageNs"]) / 1e3:8.1f}')
P
rm -rf gpurun_out/prof_t2e
