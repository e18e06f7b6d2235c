"""Diagnostic: per-tensor gradient errors of Engine.train_step against the float64 oracle at one shape (the body of
tests/test_gpu_vqvae.py::test_fused_train_step_vs_oracle without its asserts).  usage: r04_dbg.py B T D H K p"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import g2v_oracle as O
import test_gpu_vqvae as TV
B, T, D, H, K = (int(v) for v in sys.argv[1:6]); p = float(sys.argv[6])
DEV = "cuda:0"
sd = O.init_vqvae_state(D, H, 2, K, seed=3)
g = torch.Generator().manual_seed(5)
x = torch.randn(B, T, D, generator=g)
cfg = dict(n_layers=2, dropout_prob=p, commitment_cost=0.25, n_pre_poses=1, conditioned=True, w_l1=5.0, w_cont=0.1, w_var=0.5, lr=5e-4)
eng = TV._engine_from_state(sd, D, H, K, T, p)
xd = x.to(DEV)
adam = {}
sd, x = TV.as64(sd), x.double()
for step in range(2):
    masks = {"dec": (torch.rand(T - 1, B, D, generator=g) < 0.05).to(torch.uint8)}
    if p > 0:
        masks["in"] = (torch.rand(T, B, D, generator=g) < 1 - p).to(torch.uint8)
        masks["enc_l0"] = torch.ones(T, B, 2 * H, dtype=torch.uint8)
        masks["dec_l0"] = (torch.rand(T - 1, B, H, generator=g) < 1 - p).to(torch.uint8)
    with TV.default64():
        r = O.vqvae_train_step(sd, adam, x, masks, cfg)
    eng.set_masks(B, masks["dec"].to(DEV), masks["in"].to(DEV) if p > 0 else None, masks["dec_l0"].to(DEV) if p > 0 else None)
    eng.train_step(xd, xd, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5, draw_masks=False)
    b = eng.buffers(B)
    print("step", step, "y relerr", TV.relerr(b["y"].transpose(0, 1), r["outputs"]), "idx equal frac",
          float((b["idx"].cpu() == r["idx"]).float().mean()))
    for name, _ in eng.layout:
        ref = r["grads"][name]
        if float(ref.abs().max()) == 0.0:
            continue
        print(f"  {name:45s} l2 {TV.relerr_l2(eng.view(name, True), ref):.2e} max {TV.relerr(eng.view(name, True), ref):.2e}")
    TV.sync_engine_from_oracle(eng, sd, adam, step + 1)
