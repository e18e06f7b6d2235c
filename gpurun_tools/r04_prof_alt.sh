#!/bin/bash
# kernel statistics (avg / min / max) of bench.py --config native --batch 4096 for the product library and gpurun_tools/libg2v_alt.so
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for l in gesture2vec_amd/libg2v_hip.so gpurun_tools/libg2v_alt.so; do
rm -rf gpurun_out/prof_alt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_alt -- python3 gpurun_tools/bench_altlib.py $l --config native --batch 4096 --steps 20 --warmup 5 --no-cpu-baseline --no-part-d > /dev/null 2>&1 < /dev/null
f=$(ls gpurun_out/prof_alt/*/*kernel_stats.csv | head -1)
echo "== $l"
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:7]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} min {float(r["MinNs"])/1e3:8.1f} max {float(r["MaxNs"])/1e3:8.1f}')
P
done
rm -rf gpurun_out/prof_alt
