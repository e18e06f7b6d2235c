"""Bulk code assignment: exact fp32 kernel (g2v_vq_assign_fwd) vs bf16 split screening + exact re-check (g2v_vq_assign_bulk),
N sweep at E = 128, K = 512 -> profiles JSON lines.  Diagnostic / DESIGN.md."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops
dev = "cuda:0"
E, K = 128, 512
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out = []
for scale, tag in ((1.0, "N(0,1) rows and codes"), (0.05, "rows and codes scaled 0.05 (a trained model's range)"),
                   (-1.0, "N(0,1) rows and codes, 32 dead codes of 300x the norm")):
    g = torch.Generator().manual_seed(3)
    dead = scale < 0
    scale = abs(scale)
    W = torch.randn(K, E, generator=g) * scale
    if dead:
        W[64:96] *= 300.0
    W = W.to(dev)
    wsq = ops.vq_code_sqnorm(W)
    for logn in (14, 16, 18, 20):
        N = 1 << logn
        flat = (torch.randn(N, E, generator=g) * scale).to(dev)
        t_exact = timeit(lambda: ops.vq_assign(flat, flat, W, wsq, want_quantized=False))
        t_bulk = timeit(lambda: ops.vq_assign_bulk(flat, W, wsq))
        idx, und = ops.vq_assign_bulk(flat, W, wsq, want_undecided=True)
        ex = ops.vq_assign(flat, flat, W, wsq, want_quantized=False)[0]
        rec = dict(data=tag, N=N, exact_us=round(t_exact, 1), bulk_us=round(t_bulk, 1), speedup=round(t_exact / t_bulk, 2),
                   undecided_frac=round(int(und.item()) / N, 4), mismatches=int((idx != ex).sum()),
                   exact_TFs=round(2 * N * K * E / t_exact / 1e6, 1), bulk_equiv_TFs=round(2 * N * K * E / t_bulk / 1e6, 1),
                   bulk_GBps=round((N * E * 4 + N * 8) / t_bulk / 1e3, 1))
        print(json.dumps(rec)); out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r06_vq_bulk_assign_sweep.json", "w"), indent=1)
