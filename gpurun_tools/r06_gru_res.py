"""W_hh-resident GRU forward (gru_res_fwd_kernel) against the streaming kernel: bitwise equality + time per call.
usage: r06_gru_res.py [B T]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gesture2vec_amd import _lib, ops
lib = _lib.load()
DEV = "cuda:0"
H = 200


def run(B, T, lengths_on, packed, resident, reps=0, h0_on=False):
    lib.g2v_ctx_set_option(None, 4, 1 if resident else 0)
    g = torch.Generator().manual_seed(3)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(DEV)
    lens = None
    row_off = None
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
    dirs = []
    for rev in (False, True):
        dirs.append(dict(gi=r(n_rows, 3 * H) if packed else r(T, B, 3 * H), w_hh=r(3 * H, H), b_hh=r(3 * H), h0=r(B, H) if h0_on else None,
                         hs=torch.full((T, B, H), 7.0, device=DEV), h_n=torch.empty((B, H), device=DEV),
                         gates=torch.full((T, B, 4 * H), 7.0, device=DEV), reverse=rev))
    ops.gru_dirs_fwd(dirs, T, B, H, lengths=lens, row_off=row_off)
    torch.cuda.synchronize()
    dt = None
    if reps:
        for _ in range(3):
            ops.gru_dirs_fwd(dirs, T, B, H, lengths=lens, row_off=row_off)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            ops.gru_dirs_fwd(dirs, T, B, H, lengths=lens, row_off=row_off)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e6
    return dirs, lens, dt


def same(a, b, lens, T, what):
    ok = True
    for k in range(2):
        for name in ("hs", "h_n"):
            if not torch.equal(a[k][name], b[k][name]):
                ok = False
                print("  MISMATCH", what, k, name, float((a[k][name] - b[k][name]).abs().max()))
        ga, gb = a[k]["gates"], b[k]["gates"]
        if lens is not None:      # gates rows of padded positions are unwritten by both (7.0) or zero
            m = (torch.arange(T, device=DEV)[:, None] < lens[None, :])[:, :, None]
            ga, gb = torch.where(m, ga, 0), torch.where(m, gb, 0)
        if not torch.equal(ga, gb):
            ok = False
            print("  MISMATCH", what, k, "gates", float((ga - gb).abs().max()))
    return ok


cases = [(4096, 20)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for B, T in cases + [(2048, 20), (1000, 7), (4100, 5)]:
    for lengths_on, packed, h0_on in ((False, False, False), (True, False, True), (True, True, False)):
        a, lens, _ = run(B, T, lengths_on, packed, False, h0_on=h0_on)
        b, _, _ = run(B, T, lengths_on, packed, True, h0_on=h0_on)
        print(json.dumps({"B": B, "T": T, "lengths": lengths_on, "packed": packed, "h0": h0_on, "bitwise": same(a, b, lens, T, (B, T))}), flush=True)
for B, T in ((4096, 20), (2048, 20), (1024, 20), (4096, 34), (8192, 20)):
    _, _, t_s = run(B, T, False, False, False, reps=20)
    _, _, t_r = run(B, T, False, False, True, reps=20)
    print(json.dumps({"B": B, "T": T, "stream_us": round(t_s, 1), "resident_us": round(t_r, 1)}), flush=True)
