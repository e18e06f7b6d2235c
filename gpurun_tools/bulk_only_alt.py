"""bulk_only.py against another build of the library: python3 gpurun_tools/bulk_only_alt.py <lib.so> [log2 N]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gesture2vec_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from gesture2vec_amd import ops
N = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
g = torch.Generator().manual_seed(3)
W = torch.randn(512, 128, generator=g).to("cuda:0")
x = torch.randn(N, 128, generator=g).to("cuda:0")
wsq = ops.vq_code_sqnorm(W)
for _ in range(3):
    ops.vq_assign_bulk(x, W, wsq)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    ops.vq_assign_bulk(x, W, wsq)
e.record(); torch.cuda.synchronize()
print(sys.argv[1], N, round(s.elapsed_time(e) / 5 * 1e3, 1), "us per call")
