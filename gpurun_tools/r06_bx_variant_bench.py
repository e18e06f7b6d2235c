"""Round 6: one BUILD of the library (argv[1] = path of a libg2v*.so; gpurun_tools/r06_build_bx_variants.sh) -- the bf16-screened
fused VQ kernel at N = argv[2:] (default 4096) on three data sets: every output against the fp32 kernel of the same build
(bitwise), average launch time over 200 back-to-back launches (events on the launch stream), three repeats."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gesture2vec_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from gesture2vec_amd import ops
from gesture2vec_amd._lib import check
sys.path.insert(0, os.path.join(root, "gpurun_tools"))
from vq_bx_bench import data, timeit, E, K

lib = _lib.load()
Ns = [int(a) for a in sys.argv[2:]] or [4096]
for N in Ns:
    rec = {"lib": os.path.basename(sys.argv[1]), "N": N}
    for kind in ("uniform", "gru", "trained"):
        z, Wp, bp, W = data(kind, N)
        wsq = ops.vq_code_sqnorm(W)
        frag = ops.vq_pack_codebook(W)
        wpf = ops.vq_pack_codebook(Wp)
        img = ops.vq_bx_pack(W, wsq, Wp, bp)
        ref = ops.vq_fused_assign(z, Wp, bp, W, wsq, codebook_frag=frag)
        got = ops.vq_fused_assign_bx(z, wpf, bp, W, img, wsq, flags=0, want_diag=True)
        torch.cuda.synchronize()
        eq = [bool(torch.equal(a, b)) for a, b in zip(ref[:4], got[:4])]
        d = got[4].cpu().tolist()
        st = torch.cuda.current_stream().cuda_stream
        flat, idx, quant, sse = got[:4]
        a_new = (z.data_ptr(), wpf.data_ptr(), bp.data_ptr(), W.data_ptr(), img.data_ptr(), wsq.data_ptr(), flat.data_ptr(),
                 idx.data_ptr(), quant.data_ptr(), sse.data_ptr(), None, N, E, K, 0, st)
        us = [round(timeit(lambda: check(lib.g2v_vq_fused_assign_bx_fwd(*a_new))), 3) for _ in range(3)]
        fl = 2.0 * N * K * E + 2.0 * N * E * E
        rec[kind] = {"us": us, "frac": round(fl / (min(us) * 1e-6) / 157.3e12, 4), "bitwise_flat_idx_quant_sse": eq,
                     "exact_tiles": d[0], "pairs": d[1]}
    print(json.dumps(rec), flush=True)
