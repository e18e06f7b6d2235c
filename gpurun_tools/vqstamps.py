"""Diagnostic only: shader-clock stamps of the fused VQ kernel (N=4096, E=128, K=512), stamps library build."""
import ctypes, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gesture2vec_amd import _lib
_lib.LIB_PATH = os.path.join(root, "gpurun_tools", "libg2v_pstamps.so")
import torch
from gesture2vec_amd import ops
lib = _lib.load()
dev = "cuda:0"
N, E, K = 4096, 128, 512
W = (torch.rand(K, E, device=dev) * 2 - 1); wsq = ops.vq_code_sqnorm(W)
Wp = torch.randn(E, E, device=dev) * 0.1; bp = torch.randn(E, device=dev) * 0.1
z = torch.randn(N, E, device=dev)
frag = ops.vq_pack_codebook(W) if (len(sys.argv) < 2 or sys.argv[1] != "rowmajor") else None
for _ in range(10):
    ops.vq_fused_assign(z, Wp, bp, W, wsq, codebook_frag=frag)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 128)()
raw.g2v_read_vqstamps(buf)
names = ["requests+stage", "projection", "||x||^2", "xb + loop", "merge", "gather+STE"]
for b in range(4):
    st = [buf[b * 32 + k] for k in range(7)]
    print("slot", b, [st[k + 1] - st[k] for k in range(6)], "total", st[6] - st[0])
print(names)
