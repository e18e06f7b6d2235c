"""Diagnostic: train-step time of the chunk VQ-VAE at the reference's native config/VQ-VAE.yml shape (B=128, T=20, D=40,
H=200, K=512, dropout 0.2) and at the GENEA shape (T=10, D=45, H=200, K=400), generic (non-templated) kernels."""
import argparse, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
out = []
for name, (B, T, D, H, K, p) in {"native": (128, 20, 40, 200, 512, 0.2), "native_B4096": (4096, 20, 40, 200, 512, 0.2),
                                 "genea_B4096": (4096, 10, 45, 200, 400, 0.0)}.items():
    args = argparse.Namespace(rep_learning_dim=D, hidden_size=H, n_layers=2, dropout_prob=p, autoencoder_vq="True",
                              autoencoder_vae="False", autoencoder_vq_components=K, autoencoder_vq_commitment_cost=0.25,
                              autoencoder_conditioned="True", autoencoder_att="False", autoencoder_fixed_weight="False",
                              n_pre_poses=1, n_poses=T)
    torch.manual_seed(0)
    net = Autoencoder_VQVAE(args, D, T).to("cuda:0"); net.train(True)
    eng = net.engine()
    x = torch.randn(B, T, D, device="cuda:0")
    step = lambda: eng.train_step(x, x, lr=5e-4, w_l1=5, w_cont=.1, w_var=.5)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out.append(dict(cfg=name, B=B, ms_per_step=round(dt / n * 1e3, 3), chunks_per_s=round(B * n / dt, 1)))
print(json.dumps(out))
