"""Diagnostic only (never shipped / timed): s_memtime stamps (10 ns ticks) of step t = 5 of dec_cluster_fwd_kernel at the
config/VQ-VAE.yml dims, B = 128, for the workgroups (tile 0, row group 0) and (tile 5, row group 3), every wave.
Slots: 0 step start | 1 BN sums done | 2 barrier | 3 u row swept (wave 0) | 4 cell-0 products done (waves 0, 1) | 5 barrier |
6 cell-0 epilogue + publish done | 7 h0 row swept (wave 2) | 8 cell-1 products done (waves 2, 3) | 9 barrier | 10 cell-1 epilogue |
11 h1 row swept (wave 3) | 12 barrier | 13 out layer + xin | 14 barrier | 15 pre_linear + publish (wave 0)."""
import ctypes, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
src = os.path.join(root, "gesture2vec_amd", "csrc")
dbg = os.path.join(root, "gpurun_tools", "libg2v_stamps.so")
if not os.path.exists(dbg) or "--build" in sys.argv:
    srcs = subprocess.check_output(["make", "-s", "-C", src, "--eval", "print-srcs: ; @echo $(SRCS)", "print-srcs"], text=True).split()
    subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DG2V_STAMPS -shared {' '.join(srcs)} -o {dbg}", shell=True)
    if "--build" in sys.argv:
        sys.exit(0)
from gesture2vec_amd import _lib as _l0
_l0.LIB_PATH = dbg
import torch
import bench
bench.CFG.update({k: v for k, v in bench.CONFIGS["native"].items() if k != "name"})
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
B = 128
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
for wg in range(2):
    base = min(buf[(wg * 4 + w) * 16] for w in range(4))
    for w in range(4):
        st = [buf[(wg * 4 + w) * 16 + k] for k in range(16)]
        print("wg", wg, "wave", w, "ticks since step start:", [(v - base) if v else None for v in st])
d = [buf[128 + k] for k in range(16)]
print("h0 sweep of wg 0 / wave 2: start", d[0] - base0 if (base0 := min(buf[w * 16] for w in range(4))) else 0, "poll rounds", d[15],
      "round completion ticks since sweep start:", [(v - d[0]) for v in d[1:15] if v])
print("backward (dec_cluster_bwd_kernel), step t = 5: 0 top | 1 dbn swept + sums | 2 barrier | 3 du done | 4 dy stage | 5 barrier | 6 cell 1 (wave 0) | "
      "7 pair-1 products published | 8 swept | 9 barrier | 11 pair-0 products | 12 swept | 13 barrier | 14 element-wise done | 15 end")
for wg in range(2):
    base = min(buf[256 + (wg * 4 + w) * 16] for w in range(4))
    for w in range(4):
        st = [buf[256 + (wg * 4 + w) * 16 + k] for k in range(16)]
        print("bwd wg", wg, "wave", w, [(v - base) if v else None for v in st])
