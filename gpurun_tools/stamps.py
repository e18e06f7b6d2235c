"""Diagnostic only (never shipped/timed): per-phase s_memtime stamps of the decoder step kernels (fwd and bwd) at t=5."""
import ctypes, subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gesture2vec_amd", "csrc")
# every source of the library (the Makefile's SRCS), stamps compiled in, into a library of its OWN (never over the product's)
dbg = os.path.join(root, "gpurun_tools", "libg2v_stamps.so")
srcs = subprocess.check_output(["make", "-s", "-C", src, "--eval", "print-srcs: ; @echo $(SRCS)", "print-srcs"], text=True).split()
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DG2V_STAMPS -shared {' '.join(srcs)} -o {dbg}", shell=True)
from gesture2vec_amd import _lib as _l0
_l0.LIB_PATH = dbg
import torch, argparse
from gesture2vec_amd import _lib
import bench
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
lib = _lib.load()
dev = "cuda:0"
net = Autoencoder_VQVAE(bench.model_args(), 135, 34).to(dev); net.train(True)
eng = net.engine()
x = torch.randn(4096, 34, 135, device=dev)
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
    print("bwd block", b, "deltas:", [st[k + 1] - st[k] for k in range(7)], "total", st[7] - st[0], "start-vs-blk0", st[0] - buf[8], "end-vs-blk0start", st[7] - buf[8])

import numpy as np
sp = (ctypes.c_ulonglong * (1024 * 4))()
print("rc", raw.g2v_read_spans(sp))
a = np.array(list(sp), dtype=np.float64).reshape(1024, 4)[:256]
for name, c0, c1 in (("bwd", 0, 1), ("fwd", 2, 3)):
    st, en = a[:, c0], a[:, c1]
    t0 = st.min()
    dur = en - st
    print(name, "block duration ticks: min %.0f med %.0f max %.0f | start spread %.0f | last end - first start %.0f"
          % (dur.min(), np.median(dur), dur.max(), st.max() - t0, en.max() - t0))
