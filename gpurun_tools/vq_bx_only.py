"""Runs only the bf16-screened fused VQ kernel (N=4096, E=128, K=512) for rocprofv3 counter collection. argv[1] = flags (default 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops, _lib
lib = _lib.load()
dev = "cuda:0"
N, E, K = 4096, 128, 512
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W = (torch.rand(K, E, device=dev) * 2 - 1); wsq = ops.vq_code_sqnorm(W)
Wp = torch.randn(E, E, device=dev) * 0.1; bp = torch.randn(E, device=dev) * 0.1
z = torch.randn(N, E, device=dev)
wpf = ops.vq_pack_codebook(Wp); img = ops.vq_bx_pack(W, wsq, Wp, bp)
for _ in range(20):
    ops.vq_fused_assign_bx(z, wpf, bp, W, img, wsq, flags=flags)
torch.cuda.synchronize()
