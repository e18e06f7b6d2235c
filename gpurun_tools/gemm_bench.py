"""Times the dense-layer entry points at the shapes of the BASELINE step (diagnostic)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops
dev = "cuda:0"
M = 34 * 4096
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
res = {}
x64 = torch.randn(M, 64, device=dev); w192 = torch.randn(192, 64, device=dev); b192 = torch.randn(192, device=dev); y192 = torch.empty(M, 192, device=dev)
res["fwd K64 N192 (143MB)"] = timeit(lambda: ops.linear_fwd(x64, w192, b192, out=y192))
x135 = torch.randn(4096, 34, 135, device=dev); w64 = torch.randn(64, 135, device=dev); b64 = torch.randn(64, device=dev); y64 = torch.empty(M, 64, device=dev)
res["fwd K135 N64 rowmap (111MB)"] = timeit(lambda: ops.linear_fwd(x135, w64, b64, M=M, row_map=(4096, 135, 34 * 135), out=y64))
dy = torch.randn(M, 192, device=dev); dx = torch.empty(M, 64, device=dev)
res["bwd_data N192->K64 (143MB)"] = timeit(lambda: ops.linear_bwd_data(dy, w192, out=dx))
dw = torch.empty(192, 64, device=dev); db = torch.empty(192, device=dev)
res["bwd_weight N192 K64 (143MB)"] = timeit(lambda: ops.linear_bwd_weight(dy, x64, 192, 64, dw=dw, db=db))
res["bwd_weight N64 K135 rowmap (111MB)"] = timeit(lambda: ops.linear_bwd_weight(y64, x135, 64, 135, M=M, row_map=(4096, 135, 34 * 135)))
res["bwd_weight N192 K64 bf16x3"] = timeit(lambda: ops.linear_bwd_weight(dy, x64, 192, 64, dw=dw, db=db, bf16x3=True))
res["bwd_weight N64 K135 rowmap bf16x3"] = timeit(lambda: ops.linear_bwd_weight(y64, x135, 64, 135, M=M, row_map=(4096, 135, 34 * 135), bf16x3=True))
a, _ = ops.linear_bwd_weight(dy, x64, 192, 64)
b2, _ = ops.linear_bwd_weight(dy, x64, 192, 64, bf16x3=True)
ref = dy.double().t() @ x64.double()
res["relerr fp32 / bf16x3 (max-norm)"] = [float((a.double() - ref).abs().max() / ref.abs().max()), float((b2.double() - ref).abs().max() / ref.abs().max())]
t = torch.empty(M * 64, device=dev)
res["torch copy 2x143MB ref"] = timeit(lambda: y192.copy_(dy))
print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}))
