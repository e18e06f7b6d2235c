"""Diagnostic only: shader-clock stamps of the bf16-screened fused VQ kernel (N=4096, E=128, K=512); needs the stamps build
gpurun_tools/libg2v_vqstamps.so (hipcc -DG2V_VQSTAMPS of the library sources)."""
import ctypes, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gesture2vec_amd import _lib
_lib.LIB_PATH = os.path.join(root, "gpurun_tools", "libg2v_vqstamps.so")
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
names = ["z staged", "projection + sweep", "scan", "exact chains || decided rows, then the rest", "sync", "sse"]
print("flags", flags)
for b in range(4):
    st = [buf[b * 32 + k] for k in range(32)]
    print("slot", b, [st[k + 1] - st[k] for k in range(6)], "total", st[6] - st[0])
print(names)
