"""which configuration latches the fault: (B, p, chase) one at a time, fresh engines"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from oracle import g2v_oracle as O
from test_gpu_dp_engine import _engine
D, H, K, T = 135, 64, 512, 34
for B, p in ((64, 0.0), (48, 0.0), (48, 0.2), (64, 0.2), (1024, 0.2)):
    for chase in (False, True):
        for minrows in (0,):
            sd = O.init_vqvae_state(D, H, 2, K, seed=11)
            eng = _engine(sd, D, H, K, T, p)
            eng.seed = 5
            eng.loss_chase = chase
            eng.overlap_min_rows = minrows
            x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(170)).to("cuda:0")
            res = []
            for s in range(3):
                eng.train_step(x, x, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
                torch.cuda.synchronize()
                f = int(eng.lib.g2v_dec_rollout_persist_fault(1))
                res.append(f)
            print("B", B, "p", p, "chase", chase, "folded", eng.buffers(B)["loss_folded"], "fault per step", res, "loss", [round(float(v), 5) for v in eng.loss_terms.tolist()], flush=True)
