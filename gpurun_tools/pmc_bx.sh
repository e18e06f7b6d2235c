#!/bin/bash
# counter passes over the bx kernel only (20 launches per pass; one --pmc group per pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
i=0
for g in "$@"; do
  timeout 120 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcb_$i -- python3 gpurun_tools/vq_bx_only.py 0 > gpurun_out/pmcb_$i.log 2>&1
  i=$((i+1))
done
python3 - <<'P'
import csv, glob, json, collections
res = collections.OrderedDict()
for d in sorted(glob.glob("gpurun_out/pmcb_*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "vq_fused_bx_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k] = round(sum(v) / len(v), 1)
print(json.dumps(res, indent=0))
json.dump(res, open("gpurun_out/pmc_bx.json", "w"), indent=1)
P
