#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "randomized or self_check or bx" 2>&1 | tail -6
timeout 600 python bench.py > gpurun_out/r04d_bench_default.json 2> gpurun_out/r04d_bench_default.err; tail -c 1500 gpurun_out/r04d_bench_default.json; echo
tail -3 gpurun_out/r04d_bench_default.err
for cfg in native genea; do
  timeout 300 python bench.py --config $cfg --steps 100 --warmup 5 --no-cpu-baseline 2>gpurun_out/r04d_$cfg.err | tee gpurun_out/r04d_bench_$cfg.json | cut -c1-700; tail -2 gpurun_out/r04d_$cfg.err
done
timeout 300 python bench.py --config native --batch 4096 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r04d_bench_native_B4096.json | cut -c1-400
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python bench.py --force-dp --steps 200 --warmup 10 --no-cpu-baseline 2>gpurun_out/r04d_dp.err | tee gpurun_out/r04d_bench_forcedp.json | cut -c1-500; tail -3 gpurun_out/r04d_dp.err
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_t2e
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t2e -- python3 gpurun_tools/prof_t2e.py 4096 False > gpurun_out/prof_t2e.log 2>&1 < /dev/null
f=$(ls gpurun_out/prof_t2e/*/*kernel_stats.csv 2>/dev/null | head -1)
cp $f gpurun_out/r04d_kernel_stats_part_d_B4096_noatt.csv
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f'{r["Name"][:66]:66s} calls {int(r["Calls"]):6d} total_ms {float(r["TotalDurationNs"]) / 1e6:8.2f} avg_us {float(r["AverageNs"]) / 1e3:8.1f}')
P
rm -rf gpurun_out/prof_t2e
