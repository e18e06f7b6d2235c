#!/bin/bash
# kernel statistics of gpurun_tools/bench_t2e.py (Part d, B = 128 and 4096, with / without attention): top kernels by total time
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_t2e
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t2e -- python3 gpurun_tools/bench_t2e.py > gpurun_out/prof_t2e.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_t2e/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
    print(f'{r["Name"][:66]:66s} calls {int(r["Calls"]):6d} total_ms {float(r["TotalDurationNs"]) / 1e6:8.2f} avg_us {float(r["AverageNs"]) / 1e3:8.1f} max_us {float(r["MaxNs"]) / 1e3:8.1f}')
P
fi
rm -rf gpurun_out/prof_t2e
tail -1 gpurun_out/prof_t2e.log | cut -c1-900
