"""Diagnostic only (never shipped/timed): shader-clock stamps of one step (t = 10) of the persistent decoder kernels.
Loads gpurun_tools/libg2v_pstamps.so (the product sources built with -DG2V_PSTAMPS) in place of the product library."""
import ctypes, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gesture2vec_amd import _lib
_lib.LIB_PATH = os.path.join(root, "gpurun_tools", "libg2v_pstamps.so")
import torch
import bench
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
lib = _lib.load()
dev = "cuda:0"
net = Autoencoder_VQVAE(bench.model_args(), 135, 34).to(dev); net.train(True)
eng = net.engine()
x = torch.randn(4096, 34, 135, device=dev)
for _ in range(5):
    eng.train_step(x, x, lr=5e-4, w_l1=5, w_cont=.1, w_var=.5)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (2 * 4 * 24))()
print("rc", raw.g2v_read_pstamps(buf))
names = {0: ["hidden products", "exchange+stats", "BN apply", "cell0 (ih MFMA + epilogue)", "cell1", "out layer", "y/xin dense pass", "pre_linear+publish"],
         1: ["exchange + du", "dy tile stage", "feedback MFMA + epilogue", "out^T MFMA + cell1 bwd", "hh1/ih1 MFMA", "cell0 bwd", "hh0/ih0 MFMA + publish"]}
for d, nk in ((0, 9), (1, 8)):
    for b in range(4):
        st = [buf[(d * 4 + b) * 24 + k] for k in range(nk)]
        deltas = [st[k + 1] - st[k] for k in range(nk - 1)]
        print("fwd" if d == 0 else "bwd", "slot", b, "cycles:", deltas, "total", st[nk - 1] - st[0])
        if d == 0:      # extra stamps inside phase 0: after the first product, after hop 1
            x = [buf[(d * 4 + b) * 24 + k] for k in (9, 10, 11)]
            print("      phase 0 split: product 1", x[0] - st[0], "hop 1", x[1] - x[0], "product 2", st[1] - x[1],
                  "| phase 1 split: Kt + hop 2", x[2] - st[1], "stats + deferred y stores", st[2] - x[2])
    print("   phases:", names[d])
