"""Round 5: the quantiser's assignment kernel at the reference's own shapes (E = 400): generic kernel vs the packed eight-wave one,
events over 100 back-to-back launches (the codebook pack is timed separately)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import _lib, ops
lib = _lib.load()
DEV = "cuda:0"
out = []
for N, E, K in ((4096, 400, 512), (4096, 400, 400), (8192, 400, 512), (32768, 400, 512), (262144, 400, 512), (1048576, 400, 512)):
    g = torch.Generator().manual_seed(1)
    flat = torch.randn(N, E, generator=g).to(DEV)
    W = (torch.rand(K, E, generator=g) * 2 - 1).to(DEV)
    wsq = ops.vq_code_sqnorm(W)
    idx = torch.empty(N, dtype=torch.int64, device=DEV)
    frag = torch.empty(K * E, device=DEV)
    p = lambda t: t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    def timed(fn, n=100):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    n = 100 if N <= 32768 else 10
    us_pack = timed(lambda: lib.g2v_vq_pack_codebook(p(W), p(frag), K, E, st))
    us_gen = timed(lambda: lib.g2v_vq_assign_fwd(p(flat), None, p(W), p(wsq), p(idx), None, None, None, N, E, K, st), n)
    i_gen = idx.clone()
    us_pk = timed(lambda: lib.g2v_vq_assign_packed_fwd(p(flat), None, p(W), p(frag), p(wsq), p(idx), None, None, None, N, E, K, st), n)
    fl = 2.0 * N * K * E
    out.append(dict(N=N, E=E, K=K, generic_us=round(us_gen, 1), packed_us=round(us_pk, 1), pack_us=round(us_pack, 1),
                    packed_TFLOPs=round(fl / us_pk / 1e6, 1), frac_of_157=round(fl / us_pk / 1e6 / 157.3, 3),
                    hbm_GBps=round(N * (4 * E + 8) / us_pk / 1e3, 1), idx_equal=bool(torch.equal(i_gen, idx))))
    print(json.dumps(out[-1]), flush=True)
json.dump(out, open("gpurun_out/r05_vq_e400_assign.json", "w"), indent=1)
