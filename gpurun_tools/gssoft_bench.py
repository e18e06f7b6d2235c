"""train_iter_Autoencoder_VQ_seq2seq with the as-shipped soft quantiser (VQ_Payam_GSSoft) at the BASELINE shape: the engine's fused
kernel sequence (replayed from a hipGraph, what train_iter does by default) against the module-level autograd path.
usage: python gpurun_tools/gssoft_bench.py [B]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
from gesture2vec_amd.train_eval.train_seq2seq import FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev, T, D = "cuda:0", 34, 135
res = {"B": B}
for fused in (tuple(os.environ["G2V_ONLY"]) if "G2V_ONLY" in os.environ else ("1", "0")):
    os.environ["G2V_GSSOFT_FUSED"] = fused
    args = bench.model_args()
    args.autoencoder_vq_quantizer = "gssoft"
    args.loss_l1_weight, args.loss_cont_weight, args.loss_var_weight, args.learning_rate = 5.0, 0.1, 0.5, 5e-4
    torch.manual_seed(3)
    net = Autoencoder_VQVAE(args, D, T).to(dev); net.train(True)
    optim = FusedClipAdam(net, lr=5e-4, betas=(0.5, 0.999))
    x = torch.randn(B, T, D, device=dev)
    for _ in range(60):          # (the first ~50 iterations of a process run slow: code objects, clocks)
        loss, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        loss, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)       # (each iteration ends in its loss.item() sync)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    res["fused_graph" if fused == "1" else "module_autograd"] = {"ms_per_iteration": round(dt * 1e3, 4), "chunks_per_s": round(B / dt, 1),
                                                                 "loss": round(loss["loss"], 5)}
print(json.dumps(res))
