"""time per call of g2v_gru_seq_bwd (both directions, T = 20, H = 200): streaming and resident kernels, min of 3 x 30 calls (events)"""
import os, sys, json
ROOT = os.environ.get("G2V_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gesture2vec_amd import _lib, ops
lib = _lib.load()
DEV, H, T = "cuda:0", 200, 20
out = {"root": ROOT}
for B in (2048, 4096):
    g = torch.Generator().manual_seed(3)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(DEV)
    fw = [dict(gi=r(T, B, 3 * H), w_hh=r(3 * H, H), b_hh=r(3 * H), h0=None, hs=torch.empty((T, B, H), device=DEV), h_n=torch.empty((B, H), device=DEV),
               gates=torch.zeros((T, B, 4 * H), device=DEV), reverse=bool(k)) for k in range(2)]
    ops.gru_dirs_fwd(fw, T, B, H)
    dirs = [dict(d_hs=r(T, B, H), d_hn=r(B, H), hs=f["hs"], h0=None, gates=f["gates"], w_hh=f["w_hh"], dgi=torch.empty((T, B, 3 * H), device=DEV),
                 dgh=torch.empty((T, B, 3 * H), device=DEV), dh0=torch.empty((B, H), device=DEV), reverse=f["reverse"]) for f in fw]
    for name, rows in (("stream", 0), ("resident", 1025)):
        lib.g2v_ctx_set_option(None, 4, rows)
        best = 1e9
        for rep in range(3):
            for _ in range(10):
                ops.gru_dirs_bwd(dirs, T, B, H)
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(30):
                ops.gru_dirs_bwd(dirs, T, B, H)
            e1.record(st); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 30)
        out[f"B{B}_{name}_us"] = round(best, 1)
print(json.dumps(out))
