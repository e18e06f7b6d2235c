"""VQ assign kernel over an N sweep (SURVEY.md 8d caveat): achieved TFLOP/s (fp32 MFMA) and algorithmic GB/s."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops, _lib
from gesture2vec_amd._lib import check
lib = _lib.load()
dev = "cuda:0"
E, K = 128, 512
W = (torch.rand(K, E, device=dev) * 2 - 1)
wsq = ops.vq_code_sqnorm(W)
out = []
for N in (4096, 32768, 262144, 1048576):
    flat = torch.randn(N, E, device=dev); z = torch.randn(N, E, device=dev)
    idx = torch.empty(N, dtype=torch.int64, device=dev); quant = torch.empty(N, E, device=dev)
    sse = torch.empty(lib.g2v_vq_assign_blocks(N), device=dev)
    st = torch.cuda.current_stream()
    args = (flat.data_ptr(), z.data_ptr(), W.data_ptr(), wsq.data_ptr(), idx.data_ptr(), quant.data_ptr(), None, sse.data_ptr(), N, E, K, st.cuda_stream)
    reps = 200 if N <= 32768 else 20
    for _ in range(5): check(lib.g2v_vq_assign_fwd(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): check(lib.g2v_vq_assign_fwd(*args))
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * N * K * E; by = N * (8 * E + 4) + 4 * K * E
    out.append(dict(N=N, us=round(us, 2), tflops=round(fl / us / 1e6, 1), frac_mfma_f32=round(fl / us / 1e6 / 157.3, 3), alg_GBps=round(by / us / 1e3, 1)))
print(json.dumps(out))
