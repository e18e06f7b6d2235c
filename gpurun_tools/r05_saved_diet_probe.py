"""Round 5: UPPER BOUND of the saved-tensor diet of the persistent forward rollout.  The kernel stores a
saved array only when its pointer is non-NULL, so the time it spends on them can be measured by passing NULL: the rollout alone
(200 back-to-back launches, events), B = 4096, T = 34, D = 135, H = 64, with
  full       every array the backward reads (what the train step runs)
  diet       without `a` (recomputable from u + bn_stats) and `xin` (from y + the keep byte)      <- the proposed diet
  no_gates   diet + without gates0 / gates1 (NOT recomputable without a product: the bound of any diet)
  bare       y, u, h0, h1 only
Prints one JSON line; the outputs of the NULL variants are not used."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gesture2vec_amd import ops, _lib
import test_gpu_ops as TG
lib = _lib.load()
DEV = "cuda:0"
B, T, D, H = 4096, 34, 135, 64
sd = TG._dec_state(D, H, seed=21)
g = torch.Generator().manual_seed(5)
target = torch.randn(B, T, D, generator=g).to(DEV)
h_init = (torch.randn(2, B, H, generator=g) * 0.5).to(DEV)
k95 = (torch.rand(T - 1, B, D, generator=g) < 0.05).to(torch.uint8).to(DEV)
nblk = ops.dec_rollout_blocks(B)
wt, _ = TG._dec_weight_tensors(sd, DEV)
ws = ops.dec_weights_struct(wt)
full = TG._alloc_saved(T, B, D, H, nblk, DEV, 0.0)
variants = {"full": full,
            "diet": {k: (None if k in ("a", "xin") else v) for k, v in full.items()},
            "no_gates": {k: (None if k in ("a", "xin", "gates0", "gates1") else v) for k, v in full.items()},
            "bare": {k: (v if k in ("y", "u", "h0", "h1", "bn_partial", "bn_stats") else None) for k, v in full.items()}}
res = {}
for rep in range(3):
    for name, sv in variants.items():
        for _ in range(5):
            ops.dec_rollout_fwd(target, h_init, ws, sv, k95, None, 0.0, 1, True, True, T, B, D, H)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.dec_rollout_fwd(target, h_init, ws, sv, k95, None, 0.0, 1, True, True, T, B, D, H)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(round(e0.elapsed_time(e1) / 50 * 1e3, 1))
assert lib.g2v_dec_rollout_persist_fault(1) == 0
print(json.dumps({"us_per_forward_rollout_incl_pack_and_memset": res, "B": B, "T": T}))
