"""One training step's kernel timeline (start offset, duration, gap to the previous kernel) and each kernel's resources from a
rocprofv3 --kernel-trace CSV.   usage: python gpurun_tools/timeline.py <kernel_trace.csv> [anchor substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "dec_persist_fwd"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seen = {}
for r in rows:
    n = r["Kernel_Name"][:70]
    if n not in seen:
        seen[n] = 1
        print(f"{n:70s} LDS {r['LDS_Block_Size']:>6} VGPR {r['VGPR_Count']:>3} AGPR {r['Accum_VGPR_Count']:>3} WG {r['Workgroup_Size_X']:>4} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) >= 4:
    a, b = idx[-3], idx[-2]
    t0 = int(rows[a]["Start_Timestamp"])
    prev = None
    busy = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev else 0.0
        busy += e - s
        print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap {gap:6.1f} q{r.get('Queue_Id', '?'):>2} {r['Kernel_Name'][:80]}")
        prev = e
    print("period", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us; kernel time", busy / 1e3, "us")
