"""Diagnostic only: shader-clock stamps of the bf16-screened fused VQ kernel (N=4096, E=128, K=512); needs a stamps build of the
library (hipcc -DG2V_VQSTAMPS of the sources): argv[1] = flags (default 0), argv[2] = library path (default
gpurun_tools/libg2v_vqstamps.so)."""
import ctypes, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gesture2vec_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(root, "gpurun_tools", "libg2v_vqstamps.so")
import torch
from gesture2vec_amd import ops
lib = _lib.load()
dev = "cuda:0"
N, E, K = 4096, 128, 512
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W = (torch.rand(K, E, device=dev) * 2 - 1); wsq = ops.vq_code_sqnorm(W)
Wp = torch.randn(E, E, device=dev) * 0.1; bp = torch.randn(E, device=dev) * 0.1
z = torch.randn(N, E, device=dev)
wpf = ops.vq_pack_codebook(Wp); img = ops.vq_bx_pack(W, wsq, Wp, bp)
for _ in range(10):
    ops.vq_fused_assign_bx(z, wpf, bp, W, img, wsq, flags=flags)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 128)()
raw.g2v_read_vqstamps(buf)
# stamp ids in program order (thread 0 of wave 0; the same ids + 16 from thread 256 = wave 4 where that wave stamps)
order = [(0, "start"), (1, "z staged | barrier"), (7, "projection done"), (8, "sweep done"), (2, "barrier"), (9, "scan done"),
         (3, "barrier"), (10, "chains / decided rows done"), (11, "barrier"), (4, "rest finished"), (5, "syncthreads"), (6, "sse written")]
print("lib", os.path.basename(_lib.LIB_PATH), "flags", flags)
for b in range(4):
    st = [buf[b * 32 + k] for k in range(32)]
    t0 = st[0]
    w0 = " ".join(f"{name}={st[k] - t0}" for k, name in order if st[k])
    w4 = " ".join(f"[{k}]={st[16 + k] - t0}" for k in (7, 8, 9, 10) if st[16 + k])
    fine = " ".join(f"{name}={st[k] - t0}" for k, name in ((12, "scan:threshold"), (13, "scan:compared"), (14, "scan:slot returned"), (15, "scan:listed"),
                                                          (16, "chains:listed rows in LDS / start"), (17, "chains:dot product done")) if st[k])
    print(f"slot {b}: wave0: {w0}\n         wave4: {w4}\n         fine:  {fine}")
