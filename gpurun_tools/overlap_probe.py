"""Does a weight-gradient GEMM co-run with the decoder backward rollout?  Times (a) the rollout alone, (b) the GEMM alone,
(c) both on two streams.  Diagnostic only."""
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gesture2vec_amd import _lib
from gesture2vec_amd._lib import check
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
dev = "cuda:0"
CFG = bench.CFG
B, T, D = 4096, CFG["T"], CFG["D"]
torch.manual_seed(0)
net = Autoencoder_VQVAE(bench.model_args(), D, T).to(dev); net.train(True)
eng = net.engine()
x = torch.randn(B, T, D, device=dev)
kw = dict(w_l1=CFG["w_l1"], w_cont=CFG["w_cont"], w_var=CFG["w_var"], epoch=1, draw_masks=True)
for _ in range(2):
    eng.train_step(x, x, lr=5e-4, **kw)
torch.cuda.synchronize()
lib = eng.lib
b = eng.buffers(B)
H, G = eng.H, 3 * eng.H
_p = lambda t: t.data_ptr()
ws2 = torch.zeros_like(b["ws"])
M = (T - 1) * B
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gdummy = [torch.zeros(G, H, device=dev) for _ in range(4)]
bdummy = [torch.zeros(G, device=dev) for _ in range(4)]

def rollout(st):
    check(lib.g2v_dec_rollout_bwd(C.byref(eng.dec_wstruct()), C.byref(b["sv"]), C.byref(b["gr"]), _p(b["keep95"]), None, eng.p,
                                  eng.n_pre, int(eng.conditioned), T, B, D, H, _p(b["ws"]), b["ws"].numel(), st))

def gemm(st, rows=M):
    arr = (_lib.WgradItem * 4)()
    for k, (dy, xx) in enumerate(((b["dgi0"], b["a"]), (b["dgh0"], b["h0"]), (b["dgi1"], b["h0"][1:]), (b["dgh1"], b["h1"]))):
        arr[k].dy, arr[k].x, arr[k].dw, arr[k].db = _p(dy), xx.data_ptr(), _p(gdummy[k]), _p(bdummy[k])
    check(lib.g2v_linear_bwd_weight_batch(arr, 4, G, H, rows, H, G, flags, _p(ws2), ws2.numel(), st))

main = torch.cuda.current_stream(); side = torch.cuda.Stream()
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps): fn()
    e1.record(main); e1.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / reps, 1)

def both():
    side.wait_stream(main)
    gemm(side.cuda_stream)
    rollout(main.cuda_stream)
    main.wait_stream(side)

res = {"flags": flags}
res["rollout_us"] = timed(lambda: rollout(main.cuda_stream))
res["gemm_us"] = timed(lambda: gemm(main.cuda_stream))
res["gemm_wave_us"] = 0
f0 = flags; flags = 0
res["gemm_wave_us"] = timed(lambda: gemm(main.cuda_stream))
flags = f0
res["both_us"] = timed(both)
# inside the concurrent run: how long does each side take?
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
torch.cuda.synchronize()
side.wait_stream(main)
ev[0].record(side); gemm(side.cuda_stream); ev[1].record(side)
ev[2].record(main); rollout(main.cuda_stream); ev[3].record(main)
main.wait_stream(side); torch.cuda.synchronize()
res["both_gemm_us"] = round(ev[0].elapsed_time(ev[1]) * 1e3, 1)
res["both_rollout_us"] = round(ev[2].elapsed_time(ev[3]) * 1e3, 1)
print(json.dumps(res))
