"""time per call of the resident GRU forward of a library variant (G2V_LIB=path): B = 2048 (one round of workgroups), 4096"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gesture2vec_amd import _lib
if os.environ.get("G2V_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["G2V_LIB"])
from gesture2vec_amd import ops
lib = _lib.load()
DEV, H = "cuda:0", 200
out = {"lib": os.environ.get("G2V_LIB", "product")}
for B, T in [tuple(int(v) for v in a.split('x')) for a in os.environ.get('SHAPES', '2048x20,4096x20').split(',')]:
    for resident in (1, 0):
        lib.g2v_ctx_set_option(None, 4, resident)
        g = torch.Generator().manual_seed(3)
        r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(DEV)
        dirs = [dict(gi=r(T, B, 3 * H), w_hh=r(3 * H, H), b_hh=r(3 * H), h0=None, hs=torch.empty((T, B, H), device=DEV),
                     h_n=torch.empty((B, H), device=DEV), gates=torch.empty((T, B, 4 * H), device=DEV), reverse=rev) for rev in (False, True)]
        for _ in range(3):
            ops.gru_dirs_fwd(dirs, T, B, H)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ops.gru_dirs_fwd(dirs, T, B, H)
        torch.cuda.synchronize()
        out[f"B{B}_{'res' if resident else 'stream'}_us"] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
print(json.dumps(out))
