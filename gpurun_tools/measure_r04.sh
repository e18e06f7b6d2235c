#!/bin/bash
# round-4 measurements: full GPU test-suite, default bench line (incl. cpu_baseline + Part d), kernel stats + one-step timeline of
# the same command, the A/B of the loss chaser, the other configs / batch sizes, the DP form
tag=${1:-a}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r04_${tag}_gpu_tests.log; cat gpurun_out/r04_${tag}_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r04_${tag}_bench_default.json 2> gpurun_out/r04_${tag}_bench_default.err
python - <<P
import json
d = json.loads(open("gpurun_out/r04_${tag}_bench_default.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["avg_us"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["fused_rnn"]["value"])
print([ (r["att"], r["B"], r["ms_per_step"]) for r in d["text2embedding"]["runs"]], d["text2embedding"].get("cpu_baseline"))
P
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r4${tag} -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-part-d > gpurun_out/prof_r4${tag}.log 2>&1
f=$(ls gpurun_out/prof_r4${tag}/*/*kernel_trace.csv | head -1); python gpurun_tools/timeline.py $f > gpurun_out/r04_${tag}_step_timeline.txt; tail -3 gpurun_out/r04_${tag}_step_timeline.txt
cp $(ls gpurun_out/prof_r4${tag}/*/*kernel_stats.csv | head -1) gpurun_out/r04_${tag}_kernel_stats_bench_steps30.csv
rm -rf gpurun_out/prof_r4${tag}
: > gpurun_out/r04_${tag}_bench_variants.jsonl
for args in "--steps 300 --warmup 10" "--steps 300 --warmup 10 --no-loss-chase" "--steps 300 --warmup 10" "--steps 300 --warmup 10 --no-loss-chase" \
            "--steps 200 --warmup 10 --force-dp" "--steps 200 --warmup 10 --no-graph" "--steps 200 --warmup 10 --dropout 0.2" \
            "--batch 128 --steps 300" "--batch 1024 --steps 300" "--batch 2048 --steps 300" "--batch 4100 --steps 50" "--batch 8192 --steps 50" "--batch 12288 --steps 30" \
            "--config native --steps 200" "--config native --batch 4096 --steps 50" "--config genea --steps 50" "--config genea --batch 128 --steps 200"; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python bench.py --no-cpu-baseline --no-part-d $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'; keep['launch'] = d['config']['launch']; keep['custom_loss'] = d['config']['custom_loss'][:30]
keep['whole_step_frac'] = d['roofline']['whole_step']['frac']; keep['vq_us'] = d['roofline']['avg_us']
print(json.dumps(keep))" | tee -a gpurun_out/r04_${tag}_bench_variants.jsonl
done
# the as-shipped soft-quantiser model through train_iter (graph replay), and its one-step timeline
for b in 4096 128; do G2V_ONLY=1 timeout 300 python gpurun_tools/gssoft_bench.py $b 2>/dev/null | tail -1; done | tee gpurun_out/r04_${tag}_gssoft_bench.jsonl
bash gpurun_tools/gssoft_tl.sh 4096 > /dev/null; cp gpurun_out/gssoft_tl_B4096.txt gpurun_out/r04_${tag}_gssoft_step_timeline_B4096.txt; tail -1 gpurun_out/gssoft_tl_B4096.txt
# kernel statistics of the other workloads: the yml's own dims at the bench batch, two row tiles per workgroup, Part d
bash gpurun_tools/r04_prof_cfg.sh native 4096 | head -12; mv gpurun_out/r04_kernel_stats_native_B4096.csv gpurun_out/r04_${tag}_kernel_stats_native_B4096.csv
bash gpurun_tools/r04_prof_cfg.sh full 8192 | head -8; mv gpurun_out/r04_kernel_stats_full_B8192.csv gpurun_out/r04_${tag}_kernel_stats_full_B8192.csv
bash gpurun_tools/r04_prof_t2e.sh 4096 False | head -10; mv gpurun_out/r04_e_kernel_stats_part_d_B4096_attFalse.csv gpurun_out/r04_${tag}_kernel_stats_part_d_B4096_noatt.csv
