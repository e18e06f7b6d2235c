"""four weight-gradient products of one shape in one call: usage python gpurun_tools/wgrad_batch_bench.py M N K
(G2V_SMALLM_WGRAD_RT=0: the 1 x 1 tile kernel).  Prints time, error against float64, and a checksum for bitwise A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
NP = int(sys.argv[4]) if len(sys.argv) > 4 else 4
g = torch.Generator().manual_seed(1)
items = []
for p in range(NP):
    dy = torch.randn(M, N, generator=g).to("cuda:0"); x = torch.randn(M, K, generator=g).to("cuda:0")
    items.append((dy, x, torch.zeros(N, K, device="cuda:0"), torch.zeros(N, device="cuda:0")))
for _ in range(5):
    ops.linear_bwd_weight_batch(items, N, K, M=M)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.linear_bwd_weight_batch(items, N, K, M=M)
e1.record(); torch.cuda.synchronize()
err = max(float(((dw.double() - dy.double().t() @ x.double()).abs().max()) / (dy.double().t() @ x.double()).abs().max()) for dy, x, dw, db in items)
errb = max(float((db.double() - dy.double().sum(0)).abs().max()) for dy, x, dw, db in items)
chk = float(sum(dw.double().sum() + db.double().sum() for _, _, dw, db in items))
print("M N K", M, N, K, "rt", "1 (2 x 2 tiles)", "us per", NP, "products", round(e0.elapsed_time(e1) / 50 * 1e3, 1),
      "rel err", err, "bias err", errb, "checksum", repr(chk))
