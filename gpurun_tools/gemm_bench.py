"""Dense-layer forward / data-gradient timing by shape, wave-per-tile kernel vs the LDS-tiled one (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops, _lib
lib = _lib.load()
dev = "cuda:0"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, K, N) in [(128, 200, 600), (640, 200, 600), (640, 200, 512), (2560, 300, 600), (4096, 200, 600), (4096, 200, 512), (4096, 400, 200),
                  (20480, 200, 600), (20480, 200, 512), (81920, 300, 600), (81920, 200, 600)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev)
    res = []
    for rows in (0, 1 << 30):
        lib.g2v_linear_set_smallm_rows(rows)
        tf = timeit(lambda: ops.linear_fwd(x, w, b))
        tb = timeit(lambda: ops.linear_bwd_data(dy, w))
        res.append((tf, tb))
    gf = 2 * M * K * N / 1e3
    print(f"M={M:6d} K={K} N={N}  fwd lds {res[0][0]:8.1f} us ({gf/res[0][0]/1e3:6.1f} TF) wave {res[1][0]:8.1f} us ({gf/res[1][0]/1e3:6.1f} TF) | bwd_data lds {res[0][1]:8.1f} wave {res[1][1]:8.1f} us ({gf/res[1][1]/1e3:6.1f} TF)")
lib.g2v_linear_set_smallm_rows(1024)
