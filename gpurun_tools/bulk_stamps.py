"""Diagnostic: s_memtime stamps of vq_bulk_sweep_kernel (one wave of workgroups 0, 5, 128, 255) at 2^20 rows; needs a -DG2V_VQSTAMPS
build of the library: python3 gpurun_tools/bulk_stamps.py <lib.so>"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gesture2vec_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from gesture2vec_amd import ops
N = 1 << 20
g = torch.Generator().manual_seed(3)
W = torch.randn(512, 128, generator=g).to("cuda:0")
x = torch.randn(N, 128, generator=g).to("cuda:0")
wsq = ops.vq_code_sqnorm(W)
for _ in range(3):
    ops.vq_assign_bulk(x, W, wsq)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 128)()
raw.g2v_read_vqstamps(buf)
for b in range(4):
    st = [buf[b * 32 + k] for k in range(16)]
    print("slot", b, "prologue", st[1] - st[0], "first fill+barrier", st[2] - st[1],
          "chunks (compute, wait+barrier):", [(st[3 + 2 * c] - st[2 + 2 * c], st[4 + 2 * c] - st[3 + 2 * c]) for c in range(6)], "total", st[15] - st[0])
