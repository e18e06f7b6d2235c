"""weight-gradient product dW = dy^T x at one shape: usage python gpurun_tools/wgrad_shape_bench.py M N K"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
dy = torch.randn(M, N, device="cuda:0"); x = torch.randn(M, K, device="cuda:0")
for _ in range(5):
    dw, db = ops.linear_bwd_weight(dy, x, N, K, want_bias=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    dw, db = ops.linear_bwd_weight(dy, x, N, K, want_bias=True)
e1.record(); torch.cuda.synchronize()
ref = dy.double().t() @ x.double()
print("M N K", M, N, K, "rows threshold", "4095", "us per product", round(e0.elapsed_time(e1) / 50 * 1e3, 1),
      "rel err", float((dw.double() - ref).abs().max() / ref.abs().max()))
