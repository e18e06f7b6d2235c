"""Kernel breakdown of ONE hipGraph replay of the Part d train iteration from a rocprofv3 kernel trace of bench_t2e.py
(first configuration: att False, B = 128).  usage: python gpurun_tools/t2e_breakdown.py <kernel_trace.csv> [config index]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
per_cfg = len(ends) // 4
a, b = ends[cfg * per_cfg + per_cfg - 3], ends[cfg * per_cfg + per_cfg - 2]          # a late (graph-replay) iteration
seg = rows[a + 1:b + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
print("kernels", len(seg), "span us", (t1 - t0) / 1e3, "busy", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3)
c, d = collections.Counter(), collections.Counter()
for r in seg:
    n = r["Kernel_Name"][:90]
    c[n] += 1
    d[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, t in d.most_common(45):
    print(f"{n:90s} {c[n]:4d} {t:8.1f}")
if len(sys.argv) > 3:
    prev = None
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap {((s - prev) / 1e3 if prev else 0):6.1f} {r['Kernel_Name'][:100]}")
        prev = e
