#!/bin/bash
# kernel durations of g2v_vq_assign_bulk at 2^$1 rows, library $2 (default: the product's)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
lib=${2:-gesture2vec_amd/libg2v_hip.so}
rm -rf gpurun_out/prof_bulk
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bulk -- python3 gpurun_tools/bulk_only_alt.py $lib ${1:-20} > gpurun_out/prof_bulk.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_bulk/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" "$lib" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:6]:
    print(f'{sys.argv[2][-16:]} {r["Name"][:60]:60s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"]) / 1e3:9.1f}')
P
rm -rf gpurun_out/prof_bulk
