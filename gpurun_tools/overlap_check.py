"""Reads a rocprofv3 kernel trace CSV and reports how much of gru_bwd's time overlaps with gemm_tn_wave kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ks.sort()
gb = [k for k in ks if "gru_bwd" in k[2]]
tn = [k for k in ks if "gemm_tn_wave" in k[2]]
print("gru_bwd", len(gb), "tn", len(tn), "queues", sorted({k[3] for k in ks}), "streams", sorted({k[4] for k in ks}))
g = gb[-1]
print("last gru_bwd", g[0], g[1], (g[1] - g[0]) / 1e3, "us", "queue", g[3])
for t in tn:
    ov = min(g[1], t[1]) - max(g[0], t[0])
    if ov > 0 or abs(t[0] - g[0]) < 2e6:
        print("  tn", (t[0] - g[0]) / 1e3, (t[1] - g[0]) / 1e3, "dur", (t[1] - t[0]) / 1e3, "overlap", max(ov, 0) / 1e3, "queue", t[3], t[2][:60])
