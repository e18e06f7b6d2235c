"""train_iter_Autoencoder_VQ_seq2seq at the reference's own config/VQ-VAE.yml shape (B=128, T=20, D=40, H=200, K=512, dropout 0.2)
and at the BASELINE shape with B=128: ms per iteration incl. its loss.item() sync.  G2V_TRAIN_ITER_GRAPH_MIN_ROWS selects from
which batch size the iteration is replayed from a hipGraph."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
from gesture2vec_amd.train_eval.train_seq2seq import FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq
out = {"graph_min_rows": os.environ.get("G2V_TRAIN_ITER_GRAPH_MIN_ROWS", "0")}
for name, (B, T, D, H, K, p) in {"native_yml_B128": (128, 20, 40, 200, 512, 0.2), "baseline_shape_B128": (128, 34, 135, 64, 512, 0.0)}.items():
    args = argparse.Namespace(rep_learning_dim=D, hidden_size=H, n_layers=2, dropout_prob=p, autoencoder_vq="True",
                              autoencoder_vae="False", autoencoder_vq_components=K, autoencoder_vq_commitment_cost=0.25,
                              autoencoder_conditioned="True", autoencoder_att="False", autoencoder_fixed_weight="False",
                              n_pre_poses=1, n_poses=T, loss_l1_weight=5.0, loss_cont_weight=0.1, loss_var_weight=0.5, learning_rate=5e-4)
    torch.manual_seed(0)
    net = Autoencoder_VQVAE(args, D, T).to("cuda:0"); net.train(True)
    optim = FusedClipAdam(net, lr=5e-4, betas=(0.5, 0.999))
    x = torch.randn(B, T, D, device="cuda:0")
    for _ in range(60):
        loss, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 200
    for _ in range(n):
        loss, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    out[name] = {"ms_per_iteration": round(dt * 1e3, 4), "chunks_per_s": round(B / dt, 1), "loss": round(loss["loss"], 5)}
print(json.dumps(out))
