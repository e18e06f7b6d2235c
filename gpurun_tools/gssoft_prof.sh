#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for m in 1 0; do
  export G2V_ONLY=$m
  rm -rf gpurun_out/prof_gs$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gs$m -- python3 gpurun_tools/gssoft_bench.py 128 > gpurun_out/prof_gs$m.log 2>&1 < /dev/null
  f=$(ls gpurun_out/prof_gs$m/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== mode $m"; if [ -n "$f" ]; then python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernels launched", calls, "total kernel time ms", round(tot / 1e6, 2), "per iteration (55 iterations):", round(calls / 55, 1), "launches,", round(tot / 55 / 1e3, 1), "us")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6} total_us {float(r["TotalDurationNs"]) / 1e3:10.1f}')
P
  fi
  rm -rf gpurun_out/prof_gs$m
done
