"""W_hh-resident GRU BPTT (gru_res_bwd_kernel) against the streaming kernel: equality to summation order + time per call"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gesture2vec_amd import _lib, ops
lib = _lib.load()
DEV, H = "cuda:0", 200


def setup(B, T, lengths_on, packed, h0_on, dhs_on=True):
    g = torch.Generator().manual_seed(3)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(DEV)
    lens = row_off = None
    n_rows = T * B
    if lengths_on:
        lens_h = torch.sort(torch.randint(max(1, T // 5), T + 1, (B,), generator=g), descending=True).values
        lens_h[0] = T
        lens = lens_h.to(torch.int32).to(DEV)
        if packed:
            n_t = [int((lens_h > t).sum()) for t in range(T)]
            row_off = [0] * T
            for t in range(1, T):
                row_off[t] = row_off[t - 1] + n_t[t - 1]
            n_rows = sum(n_t)
    fw = []
    for rev in (False, True):
        fw.append(dict(gi=r(n_rows, 3 * H) if packed else r(T, B, 3 * H), w_hh=r(3 * H, H), b_hh=r(3 * H), h0=r(B, H) if h0_on else None,
                       hs=torch.empty((T, B, H), device=DEV), h_n=torch.empty((B, H), device=DEV),
                       gates=torch.zeros((T, B, 4 * H), device=DEV), reverse=rev))
    ops.gru_dirs_fwd(fw, T, B, H, lengths=lens, row_off=row_off)
    ups = [(r(T, B, H) if dhs_on else None, r(B, H)) for _ in range(2)]
    return fw, ups, lens, row_off, n_rows


def bwd(fw, ups, lens, row_off, n_rows, B, T, resident, reps=0):
    lib.g2v_ctx_set_option(None, 4, 1 if resident else 0)
    dirs = []
    for f, (d_hs, d_hn) in zip(fw, ups):
        dirs.append(dict(d_hs=d_hs, d_hn=d_hn, hs=f["hs"], h0=f["h0"], gates=f["gates"], w_hh=f["w_hh"],
                         dgi=torch.full((n_rows, 3 * H), 7.0, device=DEV) if row_off is not None else torch.full((T, B, 3 * H), 7.0, device=DEV),
                         dgh=torch.full((T, B, 3 * H), 7.0, device=DEV), dh0=torch.empty((B, H), device=DEV), reverse=f["reverse"]))
    ops.gru_dirs_bwd(dirs, T, B, H, lengths=lens, row_off=row_off)
    torch.cuda.synchronize()
    dt = None
    if reps:
        for _ in range(3):
            ops.gru_dirs_bwd(dirs, T, B, H, lengths=lens, row_off=row_off)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            ops.gru_dirs_bwd(dirs, T, B, H, lengths=lens, row_off=row_off)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e6
    return dirs, dt


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


for B, T in ((4096, 20), (2048, 20), (1100, 7), (4100, 5)):
    for lengths_on, packed, h0_on in ((False, False, False), (True, False, True), (True, True, False)):
        fw, ups, lens, row_off, n_rows = setup(B, T, lengths_on, packed, h0_on)
        a, _ = bwd(fw, ups, lens, row_off, n_rows, B, T, False)
        b, _ = bwd(fw, ups, lens, row_off, n_rows, B, T, True)
        errs = {n: max(rel(b[k][n], a[k][n]) for k in range(2)) for n in ("dgi", "dgh", "dh0")}
        print(json.dumps({"B": B, "T": T, "lengths": lengths_on, "packed": packed, "h0": h0_on, **{k: f"{v:.2e}" for k, v in errs.items()}}), flush=True)
for B, T in ((4096, 20), (2048, 20), (1280, 20), (4096, 34)):
    fw, ups, lens, row_off, n_rows = setup(B, T, False, False, False)
    _, t_s = bwd(fw, ups, lens, row_off, n_rows, B, T, False, reps=20)
    _, t_r = bwd(fw, ups, lens, row_off, n_rows, B, T, True, reps=20)
    print(json.dumps({"B": B, "T": T, "stream_us": round(t_s, 1), "resident_us": round(t_r, 1)}), flush=True)
