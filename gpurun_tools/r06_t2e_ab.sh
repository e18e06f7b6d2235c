#!/bin/bash
# same-box A/B of Part d (gpurun_tools/bench_t2e.py): ab_old/ (a git archive of the previous commit, built) against the tree
cd "${GRAFT_REPO_ROOT:?}"
for r in 1 2; do
  (cd ab_old && timeout 600 python gpurun_tools/bench_t2e.py 2>/dev/null | sed "s/^/old $r /")
  timeout 600 python gpurun_tools/bench_t2e.py 2>/dev/null | sed "s/^/new $r /"
done > gpurun_out/r06_g_t2e_ab.log
python3 - <<'P'
import json
for line in open("gpurun_out/r06_g_t2e_ab.log"):
    tag, r, js = line.split(" ", 2)
    print(tag, r, [(d["att"], d["B"], d.get("graph_ms")) for d in json.loads(js)])
P
