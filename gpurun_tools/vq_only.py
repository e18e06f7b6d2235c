"""Runs only the VQ assign kernel at the benchmark size (N=4096,E=128,K=512) for PMC collection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops, _lib
from gesture2vec_amd._lib import check
lib = _lib.load()
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
E, K = 128, 512
W = (torch.rand(K, E, device=dev) * 2 - 1); wsq = ops.vq_code_sqnorm(W)
flat = torch.randn(N, E, device=dev); z = torch.randn(N, E, device=dev)
idx = torch.empty(N, dtype=torch.int64, device=dev); quant = torch.empty(N, E, device=dev)
sse = torch.empty(lib.g2v_vq_assign_blocks(N), device=dev)
st = torch.cuda.current_stream()
for _ in range(20):
    check(lib.g2v_vq_assign_fwd(flat.data_ptr(), z.data_ptr(), W.data_ptr(), wsq.data_ptr(), idx.data_ptr(), quant.data_ptr(), None, sse.data_ptr(), N, E, K, st.cuda_stream))
torch.cuda.synchronize()
