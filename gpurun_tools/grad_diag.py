"""Fused train step at B=4096 vs the float64 oracle: relative error of every gradient tensor, persistent on / off."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from oracle import g2v_oracle as O
from _f64 import as64, default64
from gesture2vec_amd import _lib
import test_gpu_vqvae as TV
lib = _lib.load()
DEV = "cuda:0"
B, T, D, H, K, p = int(sys.argv[1]), 34, 135, 64, 512, float(sys.argv[2])
sd = O.init_vqvae_state(D, H, 2, K, seed=3)
g = torch.Generator().manual_seed(5)
x = torch.randn(B, T, D, generator=g)
cfg = dict(n_layers=2, dropout_prob=p, commitment_cost=0.25, n_pre_poses=1, conditioned=True, w_l1=5.0, w_cont=0.1, w_var=0.5, lr=5e-4)
masks = {"dec": (torch.rand(T - 1, B, D, generator=g) < 0.05).to(torch.uint8)}
if p > 0:
    masks["in"] = (torch.rand(T, B, D, generator=g) < 1 - p).to(torch.uint8)
    masks["enc_l0"] = torch.ones(T, B, 2 * H, dtype=torch.uint8)
    masks["dec_l0"] = (torch.rand(T - 1, B, H, generator=g) < 1 - p).to(torch.uint8)
with default64():
    r = O.vqvae_train_step(as64(sd), {}, x.double(), masks, cfg)
for persist in (1, 0):
    lib.g2v_dec_rollout_set_persistent(persist)
    eng = TV._engine_from_state(sd, D, H, K, T, p)
    eng.set_masks(B, masks["dec"].to(DEV), masks["in"].to(DEV) if p > 0 else None, masks["dec_l0"].to(DEV) if p > 0 else None)
    xd = x.to(DEV)
    eng.train_step(xd, xd, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5, draw_masks=False)
    torch.cuda.synchronize()
    b = eng.buffers(B)
    print(f"== persistent {persist}: y relerr {TV.relerr(b['y'].transpose(0,1), r['outputs']):.2e}  idx mismatches {(b['idx'].cpu() != r['idx']).sum().item()}")
    for name, _ in eng.layout:
        ref = r["grads"][name]
        if float(ref.abs().max()) == 0: continue
        got = eng.view(name, True)
        d = (got.cpu().double() - ref).abs()
        print(f"   {name:45s} relerr {float(d.max()) / float(ref.abs().max()):.2e}  (#elements off by >1e-3 of max: {(d > 1e-3 * ref.abs().max()).sum().item()} of {d.numel()})")
