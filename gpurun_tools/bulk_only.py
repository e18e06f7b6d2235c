"""g2v_vq_assign_bulk alone at N = 2^20 (rocprofv3 --pmc target).  usage: python3 gpurun_tools/bulk_only.py [log2 N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops
N = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
g = torch.Generator().manual_seed(3)
W = torch.randn(512, 128, generator=g).to("cuda:0")
x = torch.randn(N, 128, generator=g).to("cuda:0")
wsq = ops.vq_code_sqnorm(W)
for _ in range(3):
    ops.vq_assign_bulk(x, W, wsq)
torch.cuda.synchronize()
