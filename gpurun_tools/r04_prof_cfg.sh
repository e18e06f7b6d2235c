#!/bin/bash
# kernel statistics of bench.py --config $1 [--batch $2]: top kernels by total time (30 steps)
cfg=${1:-native}; B=${2:-}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_cfg
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg -- python3 bench.py --config $cfg ${B:+--batch $B} --steps 30 --warmup 5 --no-cpu-baseline --no-part-d > gpurun_out/prof_cfg.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_cfg/*/*kernel_stats.csv 2>/dev/null | head -1)
cp $f gpurun_out/r04_kernel_stats_${cfg}_B${B:-default}.csv
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} total_ms {float(r["TotalDurationNs"]) / 1e6:8.2f} avg_us {float(r["AverageNs"]) / 1e3:8.1f}')
P
rm -rf gpurun_out/prof_cfg
tail -1 gpurun_out/prof_cfg.log | cut -c1-200
