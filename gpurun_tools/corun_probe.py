"""Diagnostic: does a small kernel / a weight-gradient product on a side stream make progress beside the encoder BPTT
(gru_bwd_fast_kernel)?  Eager launches, events on both streams."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
from gesture2vec_amd._lib import check
dev = "cuda:0"
net = Autoencoder_VQVAE(bench.model_args(), 135, 34).to(dev); net.train(True)
eng = net.engine()
B = 4096
x = torch.randn(B, 34, 135, device=dev)
for _ in range(3):
    eng.train_step(x, x, lr=5e-4, w_l1=5, w_cont=.1, w_var=.5)
torch.cuda.synchronize()
b = eng.buffers(B)
side = torch.cuda.Stream()
buf = torch.zeros(100000, device=dev)
def ev(): return torch.cuda.Event(enable_timing=True)
def run(side_fn, label):
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1, s0, s1, sa0, sa1 = ev(), ev(), ev(), ev(), ev(), ev()
        # side work alone
        with torch.cuda.stream(side):
            sa0.record(); side_fn(); sa1.record()
        torch.cuda.synchronize()
        e0.record()
        eng.backward_encoder(x, B)          # main stream: BPTT + its weight gradients
        e1.record()
        side.wait_event(e0)
        with torch.cuda.stream(side):
            s0.record(); side_fn(); s1.record()
        torch.cuda.synchronize()
        res.append((round(sa0.elapsed_time(sa1) * 1e3, 1), round(e0.elapsed_time(e1) * 1e3, 1), round(e0.elapsed_time(s0) * 1e3, 1), round(e0.elapsed_time(s1) * 1e3, 1)))
    print(label, "[side alone us, main us, side start, side end (from main start)]", res)
def fill():
    buf.fill_(1.0)
def wgrad():
    with torch.cuda.stream(side):
        pass
    eng_stream = torch.cuda.current_stream()
    wg, wg4 = eng._wgrad_fns(b, 33 * B, "ws_dec_wgrad")
    pre = "decoder.decoder."
    wg(b["du"].data_ptr(), 64, b["dec_xin"].data_ptr(), 135, pre + "pre_linear.0.weight", pre + "pre_linear.0.bias", 64, 135)
run(fill, "fill 100k floats")
run(wgrad, "pre_linear weight gradient")
