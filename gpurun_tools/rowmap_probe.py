"""Does the (B,T,D) -> (T,B,D) row map cost the encoder's input layer (gemm_nt_k4_kernel) and its weight gradient HBM efficiency?
Same kernels, same sizes, x read through the row map (rows 18 KB apart) against x already in (T,B,D) order (rows adjacent)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gesture2vec_amd import ops
B, T, D, H = 4096, 34, 135, 64
dev = "cuda:0"
x_btd = torch.randn(B, T, D, device=dev)
x_tbd = x_btd.transpose(0, 1).contiguous()
w = torch.randn(H, D, device=dev) * 0.1
bias = torch.randn(H, device=dev)
out = torch.empty(T * B, H, device=dev)
dy = torch.randn(T * B, H, device=dev)
dy2 = torch.randn(T * B, H, device=dev)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = timeit(lambda: ops.linear_fwd(x_btd, w, bias, M=T * B, ldx=D, row_map=(B, D, T * D), out=out))
ya = out.clone()
b = timeit(lambda: ops.linear_fwd(x_tbd, w, bias, M=T * B, out=out))
print(f"input layer forward: through the row map {a:.1f} us, rows adjacent {b:.1f} us, max diff {float((ya - out).abs().max()):.2e}")
a = timeit(lambda: ops.linear_bwd_weight_sum2(dy, dy2, x_btd, H, D, M=T * B, row_map=(B, D, T * D)))
b = timeit(lambda: ops.linear_bwd_weight_sum2(dy, dy2, x_tbd, H, D, M=T * B))
print(f"input layer weight gradient (two addends): through the row map {a:.1f} us, rows adjacent {b:.1f} us")
