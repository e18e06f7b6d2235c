"""native VQ-VAE.yml dims at a given batch size: ms per train step, hipGraph replay (diagnostic)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
B, T, D, H, K, p = int(sys.argv[1]), 20, 40, 200, 512, 0.2
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
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): step()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print(B, "ms_per_step", round((time.perf_counter() - t0) / 20 * 1e3, 3))
