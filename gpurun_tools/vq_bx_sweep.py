"""N-sweep of the product's fused VQ kernel (g2v_vq_fused_assign_bx_fwd) next to the round-2 fp32 fused kernel: average launch
time (events on the launch stream), algorithmic TFLOP/s (2NKE + 2NE^2) against the 157.3 TF fp32-MFMA peak, HBM view."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops, _lib
from gesture2vec_amd._lib import check
lib = _lib.load()
dev = "cuda:0"
E, K = 128, 512
out = {}
g = torch.Generator(device=dev).manual_seed(0)
W = torch.rand(K, E, device=dev, generator=g) * 2 - 1
Wp = torch.randn(E, E, device=dev, generator=g) * 0.1
bp = torch.randn(E, device=dev, generator=g) * 0.1
wsq = ops.vq_code_sqnorm(W); frag = ops.vq_pack_codebook(W); wpf = ops.vq_pack_codebook(Wp); img = ops.vq_bx_pack(W, wsq, Wp, bp)
st = torch.cuda.current_stream()
def timed(fn, reps):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for N in (4096, 8192, 32768, 262144, 1048576):
    z = torch.randn(N, E, device=dev, generator=g)
    flat = torch.empty(N, E, device=dev); quant = torch.empty(N, E, device=dev)
    idx = torch.empty(N, dtype=torch.int64, device=dev); sse = torch.empty(lib.g2v_vq_assign_blocks(N), device=dev)
    diag = torch.zeros(4, dtype=torch.int32, device=dev)
    a_new = lambda dg, fl: (z.data_ptr(), wpf.data_ptr(), bp.data_ptr(), W.data_ptr(), img.data_ptr(), wsq.data_ptr(), flat.data_ptr(),
                            idx.data_ptr(), quant.data_ptr(), sse.data_ptr(), dg, N, E, K, fl, st.cuda_stream)
    a_old = (z.data_ptr(), Wp.data_ptr(), bp.data_ptr(), W.data_ptr(), frag.data_ptr(), wsq.data_ptr(), flat.data_ptr(), idx.data_ptr(),
             quant.data_ptr(), sse.data_ptr(), N, E, K, st.cuda_stream)
    reps = 200 if N <= 32768 else 30
    us_new = timed(lambda: check(lib.g2v_vq_fused_assign_bx_fwd(*a_new(None, 0))), reps)
    check(lib.g2v_vq_fused_assign_bx_fwd(*a_new(diag.data_ptr(), 0))); i_new = idx.clone()
    us_old = timed(lambda: check(lib.g2v_vq_fused_assign_packed_fwd(*a_old)), reps)
    torch.cuda.synchronize()
    fl = 2.0 * N * K * E + 2.0 * N * E * E
    by = N * (12 * E + 8) + 4 * E * E + 4 * E + 4 * K * E + 4 * K
    d = diag.cpu().tolist(); tiles = (N + 15) // 16
    out[f"N={N}"] = {"bx_us": round(us_new, 2), "bx_TFLOPs": round(fl / us_new / 1e6, 1), "bx_frac_fp32_peak": round(fl / us_new / 1e6 / 157.3, 4),
                     "bx_hbm_GBps": round(by / us_new / 1e3, 1), "fp32_kernel_us": round(us_old, 2),
                     "fp32_kernel_frac": round(fl / us_old / 1e6 / 157.3, 4), "idx_equal_on_every_row": bool(torch.equal(i_new, idx)),
                     "tiles_on_exact_sweep": d[0], "pairs_per_tile": round(d[1] / max(tiles - d[0], 1), 2)}
    print(f"N={N}", json.dumps(out[f"N={N}"]), flush=True)
json.dump({"what": "fused pre_linear + assign, E=128, K=512, U(-1,1) codebook, N(0,1) rows; events on the launch stream", **out},
          open("gpurun_out/r03_vq_bx_N_sweep.json", "w"), indent=1)
