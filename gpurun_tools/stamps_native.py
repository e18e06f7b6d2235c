"""Diagnostic only (never shipped / timed): per-phase s_memtime stamps of the GENERIC decoder step kernels (dec_step_fwd/bwd_kernel<0,0>)
at the config/VQ-VAE.yml dims (H = 200, D = 40, T = 20), B = 4096, step t = 5.  Builds gpurun_tools/libg2v_stamps.so (-DG2V_STAMPS).
forward phases: 0-1 zero padding, 1-2 BatchNorm partial reduction, 2-3 a_t + stage h, 3-4 GRU cell 0, 4-5 cell 1, 5-6 out layer + y/xin,
6-7 pre_linear + partials;  backward: 0-1 BN-backward partial reduction, 1-2 du, 2-3 dy (feedback), 3-4 cell 1, 4-5 hh1/ih1 products,
5-6 cell 0, 6-7 hh0/ih0 products + dbn.  Units: s_memtime ticks (10 ns)."""
import ctypes, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
src = os.path.join(root, "gesture2vec_amd", "csrc")
dbg = os.path.join(root, "gpurun_tools", "libg2v_stamps.so")
srcs = subprocess.check_output(["make", "-s", "-C", src, "--eval", "print-srcs: ; @echo $(SRCS)", "print-srcs"], text=True).split()
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DG2V_STAMPS -shared {' '.join(srcs)} -o {dbg}", shell=True)
from gesture2vec_amd import _lib as _l0
_l0.LIB_PATH = dbg
import torch
import bench
bench.CFG.update({k: v for k, v in bench.CONFIGS["native"].items() if k != "name"})
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
net = Autoencoder_VQVAE(bench.model_args(), bench.CFG["D"], bench.CFG["T"]).to(dev); net.train(True)
eng = net.engine()
x = torch.randn(B, bench.CFG["T"], bench.CFG["D"], device=dev)
for _ in range(3):
    eng.train_step(x, x, lr=5e-4, w_l1=5, w_cont=.1, w_var=.5)
torch.cuda.synchronize()
raw = ctypes.CDLL(dbg)
buf = (ctypes.c_ulonglong * (64 * 16))()
print("rc", raw.g2v_read_stamps(buf))
for b in range(8):
    st = [buf[b * 16 + k] for k in range(8)]
    print("fwd block", b, "deltas:", [st[k + 1] - st[k] for k in range(7)], "total", st[7] - st[0])
for b in range(8):
    st = [buf[b * 16 + 8 + k] for k in range(8)]
    print("bwd block", b, "deltas:", [st[k + 1] - st[k] for k in range(7)], "total", st[7] - st[0])
