"""Persistent vs per-step decoder rollout: per-array, per-time-step max abs differences (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gesture2vec_amd import ops, _lib
import test_gpu_ops as TG
lib = _lib.load()
DEV = "cuda:0"
B, T, D, H = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 6, 135, 64
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if len(sys.argv) > 4 and sys.argv[4] == "poison":
    blocks = [torch.full((mb * 1024 * 1024 // 4,), float("nan"), device="cuda:0") for mb in (2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1) for _ in range(2)]
    blocks += [torch.full((n,), float("nan"), device="cuda:0") for n in (1 << 16, 1 << 14, 1 << 12, 1 << 10, 256, 64) for _ in range(8)]
    del blocks
sd = TG._dec_state(D, H, seed=21)
g = torch.Generator().manual_seed(5)
target = torch.randn(B, T, D, generator=g).to(DEV)
h_init = (torch.randn(2, B, H, generator=g) * 0.5).to(DEV)
k95 = (torch.rand(T - 1, B, D, generator=g) < 0.05).to(torch.uint8).to(DEV)
kl0 = (torch.rand(T - 1, B, H, generator=g) < (1 - p)).to(torch.uint8).to(DEV) if p > 0 else None
gy = (torch.randn(T, B, D, generator=g) / (T * B * D) * 100).to(DEV)
nblk = ops.dec_rollout_blocks(B)
G = 3 * H
z = lambda *s: torch.zeros(*s, device=DEV)
def run(persistent, do_bwd=True):
    prev = lib.g2v_dec_rollout_set_persistent(int(persistent))
    wt, _ = TG._dec_weight_tensors(sd, DEV)
    ws = ops.dec_weights_struct(wt)
    saved = TG._alloc_saved(T, B, D, H, nblk, DEV, p)
    ops.dec_rollout_fwd(target, h_init, ws, saved, k95, kl0, p, 1, True, True, T, B, D, H)
    torch.cuda.synchronize()
    out = {k: v.clone() for k, v in saved.items() if v is not None and k != "bn_partial"}
    if do_bwd:
        lib.g2v_dec_rollout_set_persistent(0)          # same (per-step) forward arrays feed both backward variants
        saved2 = TG._alloc_saved(T, B, D, H, nblk, DEV, p)
        ops.dec_rollout_fwd(target, h_init, ws, saved2, k95, kl0, p, 1, True, True, T, B, D, H)
        lib.g2v_dec_rollout_set_persistent(int(persistent))
        grads = {"dy": gy.clone(), "du": z(T - 1, B, H), "dbn": z(T - 1, B, H), "dgi0": z(T - 1, B, G), "dgh0": z(T - 1, B, G),
                 "dgi1": z(T - 1, B, G), "dgh1": z(T - 1, B, G), "dh_init": z(2, B, H), "d_bn_w": z(H), "d_bn_b": z(H),
                 "bn_bwd_partial": z(2, nblk, 2, H)}
        ops.dec_rollout_bwd(ws, saved2, grads, k95, kl0, p, 1, True, T, B, D, H)
        torch.cuda.synchronize()
        out.update({"g_" + k: v.clone() for k, v in grads.items() if k not in ("bn_bwd_partial", "dbn")})
    lib.g2v_dec_rollout_set_persistent(prev)
    return out
a, b = run(True), run(False)
for k in a:
    x, y = a[k], b[k]
    if x.dim() >= 2 and x.shape[0] in (T, T - 1):
        errs = [(float((x[t] - y[t]).abs().max()), float(y[t].abs().max())) for t in range(x.shape[0])]
        print(f"{k:10s}", " ".join(f"{e:.1e}/{s:.1e}" for e, s in errs[:8]), "..." if len(errs) > 8 else "")
    else:
        d = (x - y).abs()
        print(f"{k:10s} {float(d.max()):.2e} / {float(y.abs().max()):.2e}")
        if k == "g_dh_init" and float(d.max()) > 1e-3 * float(y.abs().max()):
            bad = (d > 1e-3 * float(y.abs().max())).nonzero()
            print("   bad entries (layer,row,col) first 20:", bad[:20].tolist(), "count", bad.shape[0], "rows", sorted(set((bad[:, 1] // 16).tolist()))[:20])
