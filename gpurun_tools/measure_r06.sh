#!/bin/bash
# round-6 measurements on one box: full GPU suite, default bench line, kernel stats + one-step timeline of the same command,
# FETCH / WRITE of the rollouts + the quantiser, variants, Part d kernel stats
tag=${1:-e}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r06_${tag}_gpu_tests.log; cat gpurun_out/r06_${tag}_gpu_tests.log
timeout 900 python bench.py > gpurun_out/r06_${tag}_bench_default.json 2> gpurun_out/r06_${tag}_bench_default.err
python - <<P
import json
d = json.loads(open("gpurun_out/r06_${tag}_bench_default.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d.get("sustained"), d["roofline"]["avg_us"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["functional_oracle"]["value"])
print("train_iter", d.get("train_iter")); print("bulk_assign", d.get("bulk_assign")); print("shipped", d.get("shipped_config"))
print([(r["att"], r["B"], r["ms_per_step"], r["frac_of_f32_mfma_peak"]) for r in d["text2embedding"]["runs"]], d["text2embedding"].get("cpu_baseline"))
P
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r6${tag} -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-part-d --sustained 0 > gpurun_out/prof_r6${tag}.log 2>&1
f=$(ls gpurun_out/prof_r6${tag}/*/*kernel_trace.csv | head -1); python gpurun_tools/timeline.py $f > gpurun_out/r06_${tag}_step_timeline.txt; tail -3 gpurun_out/r06_${tag}_step_timeline.txt
cp $(ls gpurun_out/prof_r6${tag}/*/*kernel_stats.csv | head -1) gpurun_out/r06_${tag}_kernel_stats_bench_steps30.csv
rm -rf gpurun_out/prof_r6${tag}
: > gpurun_out/r06_${tag}_bench_variants.jsonl
for args in "--steps 300 --warmup 10" "--steps 200 --warmup 10 --force-dp" "--steps 200 --warmup 10 --dropout 0.2" \
            "--batch 128 --steps 300" "--batch 1024 --steps 300" "--batch 2048 --steps 300" "--batch 4100 --steps 50" "--batch 8192 --steps 50" \
            "--config native --steps 200" "--config native --batch 4096 --steps 50" "--config genea --steps 50" "--config genea --batch 128 --steps 200"; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python bench.py --no-cpu-baseline --no-part-d --sustained 0 $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
keep = {k: d[k] for k in ('value', 'ms_per_step', 'steps')}
keep['args'] = '$args'; keep['launch'] = d['config']['launch']
keep['whole_step_frac'] = d['roofline']['whole_step']['frac']; keep['vq_us'] = d['roofline']['avg_us']; keep['dp_diag'] = d['config'].get('dp_diag')
print(json.dumps(keep))" | tee -a gpurun_out/r06_${tag}_bench_variants.jsonl
done
bash gpurun_tools/pmc_vq_bx.sh > /dev/null 2>&1; cp gpurun_out/r03_vqbx_pmc_traffic.json gpurun_out/r06_${tag}_vqbx_pmc_traffic.json; head -12 gpurun_out/r06_${tag}_vqbx_pmc_traffic.json
bash gpurun_tools/r04_prof_t2e.sh 4096 False | head -12; mv gpurun_out/r04_e_kernel_stats_part_d_B4096_attFalse.csv gpurun_out/r06_${tag}_kernel_stats_part_d_B4096_noatt.csv
bash gpurun_tools/r04_prof_t2e.sh 4096 True | head -8; mv gpurun_out/r04_e_kernel_stats_part_d_B4096_attTrue.csv gpurun_out/r06_${tag}_kernel_stats_part_d_B4096_att.csv
bash gpurun_tools/r04_prof_cfg.sh native 4096 | head -10; mv gpurun_out/r04_kernel_stats_native_B4096.csv gpurun_out/r06_${tag}_kernel_stats_native_B4096.csv
bash gpurun_tools/r04_prof_cfg.sh native 128 | head -14; mv gpurun_out/r04_kernel_stats_native_B128.csv gpurun_out/r06_${tag}_kernel_stats_native_B128.csv
bash gpurun_tools/r04_prof_t2e.sh 128 False | head -12; mv gpurun_out/r04_e_kernel_stats_part_d_B128_attFalse.csv gpurun_out/r06_${tag}_kernel_stats_part_d_B128_noatt.csv
bash gpurun_tools/r04_tl_cfg.sh native 128 > /dev/null 2>&1; cp gpurun_out/r04_timeline_native_B128_libg2v_hip.txt gpurun_out/r06_${tag}_timeline_native_B128.txt; tail -16 gpurun_out/r06_${tag}_timeline_native_B128.txt
python gpurun_tools/bulk_assign_bench.py 2>&1 | grep -v amdgpu.ids | grep 1048576; cp gpurun_out/r06_vq_bulk_assign_sweep.json gpurun_out/r06_${tag}_vq_bulk_assign_sweep.json
