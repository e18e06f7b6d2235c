"""Throughput of the drop-in training iteration `train_iter_Autoencoder_VQ_seq2seq` (incl. its per-iteration loss.item()) at
the reference's own config/VQ-VAE.yml shape and at the BASELINE dims.  (Replaying the small-batch iteration from a hipGraph was
measured with this script: 2.29 vs 2.26 ms native, 1.35 vs 1.36 ms at BASELINE dims B = 128 -- the ~300 dependent launches
are GPU-side latency, not host launch cost -- and not kept.)"""
import argparse, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
from gesture2vec_amd.train_eval.train_seq2seq import train_iter_Autoencoder_VQ_seq2seq, FusedClipAdam
out = []
for name, (B, T, D, H, K, p) in {"VQ-VAE.yml native": (128, 20, 40, 200, 512, 0.2), "BASELINE dims, B=128": (128, 34, 135, 64, 512, 0.0),
                                 "BASELINE dims, B=512": (512, 34, 135, 64, 512, 0.0)}.items():
    for graph in ("0",):
        args = argparse.Namespace(rep_learning_dim=D, hidden_size=H, n_layers=2, dropout_prob=p, autoencoder_vq="True",
                                  autoencoder_vae="False", autoencoder_vq_components=K, autoencoder_vq_commitment_cost=0.25,
                                  autoencoder_conditioned="True", autoencoder_att="False", autoencoder_fixed_weight="False",
                                  n_pre_poses=1, n_poses=T, loss_l1_weight=5.0, loss_cont_weight=0.1, loss_var_weight=0.5,
                                  learning_rate=5e-4)
        torch.manual_seed(0)
        net = Autoencoder_VQVAE(args, D, T).to("cuda:0"); net.train(True)
        optim = FusedClipAdam(net, 5e-4, betas=(0.5, 0.999))
        x = torch.randn(B, T, D, device="cuda:0")
        for _ in range(5): train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 100
        for _ in range(n): loss, perp = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out.append(dict(cfg=name, B=B, graph=graph == "1", ms_per_iter=round(dt / n * 1e3, 3), chunks_per_s=round(B * n / dt, 1), loss=round(loss["loss"], 5)))
        print(json.dumps(out[-1]))
