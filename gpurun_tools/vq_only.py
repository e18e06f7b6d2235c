"""Runs only the VQ kernel of the product path at the benchmark size (N=4096,E=128,K=512) for PMC collection:
argv[2] = fused (default; pre_linear + assign, g2v_vq_fused_assign_fwd) | packed (the same from the fragment image) | assign (g2v_vq_assign_fwd) | stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops, _lib
from gesture2vec_amd._lib import check
lib = _lib.load()
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
which = sys.argv[2] if len(sys.argv) > 2 else "fused"
E, K = 128, 512
W = (torch.rand(K, E, device=dev) * 2 - 1); wsq = ops.vq_code_sqnorm(W)
Wp = torch.randn(E, E, device=dev) * 0.1; bp = torch.randn(E, device=dev) * 0.1
flat = torch.randn(N, E, device=dev); z = torch.randn(N, E, device=dev)
idx = torch.randint(0, K, (N,), dtype=torch.int64, device=dev); quant = torch.empty(N, E, device=dev)
sse = torch.empty(lib.g2v_vq_assign_blocks(N), device=dev)
stats = torch.empty(K + K * E, device=dev)
ws = torch.empty(lib.g2v_vq_stats_workspace(N, E, K) + 256, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream()
for _ in range(20):
    if which == "fused":
        check(lib.g2v_vq_fused_assign_fwd(z.data_ptr(), Wp.data_ptr(), bp.data_ptr(), W.data_ptr(), wsq.data_ptr(), flat.data_ptr(),
                                          idx.data_ptr(), quant.data_ptr(), sse.data_ptr(), N, E, K, st.cuda_stream))
    elif which == "packed":
        frag = ops.vq_pack_codebook(W)
        check(lib.g2v_vq_fused_assign_packed_fwd(z.data_ptr(), Wp.data_ptr(), bp.data_ptr(), W.data_ptr(), frag.data_ptr(), wsq.data_ptr(),
                                                 flat.data_ptr(), idx.data_ptr(), quant.data_ptr(), sse.data_ptr(), N, E, K, st.cuda_stream))
    elif which == "assign":
        check(lib.g2v_vq_assign_fwd(flat.data_ptr(), z.data_ptr(), W.data_ptr(), wsq.data_ptr(), idx.data_ptr(), quant.data_ptr(), None,
                                    sse.data_ptr(), N, E, K, st.cuda_stream))
    else:
        check(lib.g2v_vq_stats(idx.data_ptr(), flat.data_ptr(), stats.data_ptr(), N, E, K, ws.data_ptr(), ws.numel(), st.cuda_stream))
torch.cuda.synchronize()
