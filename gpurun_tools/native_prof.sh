#!/bin/bash
# the reference's native config/VQ-VAE.yml shape (B=128, T=20, D=40, H=200): step time + kernel statistics of prof_native.py
cd "${GRAFT_REPO_ROOT:?}"
timeout 300 python gpurun_tools/bench_native.py 2>/dev/null < /dev/null | tail -1
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_nat
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nat -- python3 gpurun_tools/prof_native.py ${1:-128} > gpurun_out/prof_nat.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_nat/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = 10
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("per step:", round(calls / n, 1), "launches,", round(tot / n / 1e3, 1), "us of kernels")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(f'{r["Name"][:64]:64s} calls/step {int(r["Calls"]) / n:6.1f} us/step {float(r["TotalDurationNs"]) / n / 1e3:8.1f} avg_us {float(r["AverageNs"]) / 1e3:7.1f}')
P
fi
rm -rf gpurun_out/prof_nat
