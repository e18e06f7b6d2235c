#!/bin/bash
# round 6: full GPU suite + default bench + Part d A/B (ab_old = previous commit) with the side branches in
cd "${GRAFT_REPO_ROOT:?}"
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_i_pytest.log 2>&1
tail -3 gpurun_out/r06_i_pytest.log
bash gpurun_tools/r06_t2e_ab.sh
cp gpurun_out/r06_g_t2e_ab.log gpurun_out/r06_i_t2e_ab.log
timeout 900 python bench.py > gpurun_out/r06_i_bench_default.json 2> gpurun_out/r06_i_bench_default.err
echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r06_i_bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], [ (r['att'],r['B'],r['ms_per_step']) for r in d['text2embedding']['runs']], d.get('shipped_config',{}).get('ms_per_step'))"
