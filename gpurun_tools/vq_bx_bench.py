"""Round 3: the bf16-screened fused VQ kernel (g2v_vq_fused_assign_bx_fwd) against the fp32 fused kernel of round 2:
bitwise equality of every output on several data sets, share of tiles / pairs that needed the exact arithmetic, and the
average launch time of each variant (events on the launch stream, back-to-back launches).
  python gpurun_tools/vq_bx_bench.py [N ...]          (default N = 4096)
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops, _lib
from gesture2vec_amd._lib import check

lib = _lib.load()
dev = "cuda:0"
E, K = 128, 512


def data(kind, N, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    Wp = torch.randn(E, E, device=dev, generator=g) * 0.1
    bp = torch.randn(E, device=dev, generator=g) * 0.1
    z = torch.randn(N, E, device=dev, generator=g)
    if kind == "uniform":                 # the reference's initial codebook (:1204)
        W = torch.rand(K, E, device=dev, generator=g) * 2 - 1
    elif kind == "gru":                   # z like GRU states in (-1, 1), pre_linear at its nn.Linear default init
        z = torch.tanh(z)
        Wp = (torch.rand(E, E, device=dev, generator=g) * 2 - 1) / E ** 0.5
        bp = (torch.rand(E, device=dev, generator=g) * 2 - 1) / E ** 0.5
        W = torch.rand(K, E, device=dev, generator=g) * 2 - 1
    elif kind == "trained":               # codes = projected rows of the data (what the EMA update converges to) + noise
        flat = z @ Wp.t() + bp
        W = flat[torch.randint(0, N, (K,), device=dev, generator=g)] + 0.05 * torch.randn(K, E, device=dev, generator=g)
    elif kind == "collapsed":             # all codes nearly identical: every code a candidate -> exact sweep everywhere
        W = torch.randn(1, E, device=dev, generator=g).expand(K, E) + 1e-4 * torch.randn(K, E, device=dev, generator=g)
        W = W.contiguous()
    elif kind == "ties":                  # duplicated codes: exact ties, lowest index must win
        W = torch.rand(K, E, device=dev, generator=g) * 2 - 1
        W[K // 2:] = W[:K // 2]
    else:
        raise ValueError(kind)
    return z.contiguous(), Wp.contiguous(), bp.contiguous(), W.contiguous()


def timeit(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    Ns = [int(a) for a in sys.argv[1:]] or [4096]
    out = []
    for N in Ns:
        for kind in ("uniform", "gru", "trained", "ties", "collapsed"):
            z, Wp, bp, W = data(kind, N)
            wsq = ops.vq_code_sqnorm(W)
            frag = ops.vq_pack_codebook(W)
            wpf = ops.vq_pack_codebook(Wp)
            img = ops.vq_bx_pack(W, wsq, Wp, bp)
            ref = ops.vq_fused_assign(z, Wp, bp, W, wsq, codebook_frag=frag)
            rec = {"N": N, "data": kind}
            for name, flags in (("bx2", 0), ("bx_exact", 1)):
                got = ops.vq_fused_assign_bx(z, wpf, bp, W, img, wsq, flags=flags, want_diag=True)
                torch.cuda.synchronize()
                eq = [bool(torch.equal(a, b)) for a, b in zip(ref[:3], got[:3])] + [bool(torch.allclose(ref[3], got[3], rtol=2e-6, atol=0, equal_nan=True))]
                nd = int((ref[1] != got[1]).sum())
                d = got[4].cpu().tolist()
                rec[name] = {"flat_idx_quant_sse_bitwise": eq, "idx_mismatch": nd, "exact_tiles": d[0], "pairs": d[1],
                             "tiles": (N + 15) // 16}
            st = torch.cuda.current_stream().cuda_stream
            flat, idx, quant, sse = ref
            a_old = (z.data_ptr(), Wp.data_ptr(), bp.data_ptr(), W.data_ptr(), frag.data_ptr(), wsq.data_ptr(), flat.data_ptr(),
                     idx.data_ptr(), quant.data_ptr(), sse.data_ptr(), N, E, K, st)
            rec["us_fp32_packed"] = round(timeit(lambda: check(lib.g2v_vq_fused_assign_packed_fwd(*a_old))), 3)
            for name, flags in (("bx2", 0), ("bx_exact", 1)):
                a_new = (z.data_ptr(), wpf.data_ptr(), bp.data_ptr(), W.data_ptr(), img.data_ptr(), wsq.data_ptr(), flat.data_ptr(),
                         idx.data_ptr(), quant.data_ptr(), sse.data_ptr(), None, N, E, K, flags, st)
                rec["us_" + name] = round(timeit(lambda: check(lib.g2v_vq_fused_assign_bx_fwd(*a_new))), 3)
            fl = 2.0 * N * K * E + 2.0 * N * E * E
            rec["frac_fp32_peak_bx2"] = round(fl / (rec["us_bx2"] * 1e-6) / 157.3e12, 4)
            out.append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/vq_bx_bench.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
