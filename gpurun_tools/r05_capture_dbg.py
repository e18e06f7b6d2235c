"""which part of the fused Part-d rollout breaks hipGraph capture (segfault in capture_end)?  one variant per subprocess"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for v in ("fwd_small", "fwdbwd_small", "fwd_big", "fwdbwd_big", "fwdbwd_big_noattr", "bptt_only_big"):
        r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, v], capture_output=True, text=True)
        print(v, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1][:200], "|", [l for l in r.stderr.splitlines() if "Error" in l or "error" in l][:3])
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gesture2vec_amd import ops, _lib
v = sys.argv[1]
big = "big" in v
B, H, K, S1 = (4096, 200, 512, 5) if big else (64, 32, 64, 5)
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)
G = 3 * H
wd = dict(emb=r(K, H), w_pre=r(H, H), b_pre=r(H), bn_w=r(H) + 1, bn_b=r(H), bn_running_mean=torch.zeros(H, device=dev), bn_running_var=torch.ones(H, device=dev),
          w_ih0=r(G, H), w_hh0=r(G, H), b_ih0=r(G), b_hh0=r(G), w_ih1=r(G, H), w_hh1=r(G, H), b_ih1=r(G), b_hh1=r(G), w_out=r(K, H), b_out=r(K))
f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
nblk = (B + 15) // 16
sv = dict(ids=torch.empty((S1, B), dtype=torch.int64, device=dev), ec=f32(S1, B, H), u=f32(S1, B, H), a=f32(S1, B, H), bn_stats=f32(S1, 2, H),
          h0=f32(S1 + 1, B, H), h1=f32(S1 + 1, B, H), gates0=f32(S1, B, 4 * H), gates1=f32(S1, B, 4 * H), logits=f32(S1, B, K), bn_partial=f32(2, nblk, 2, H))
codes = torch.randint(0, K, (S1 + 1, B), generator=g).to(dev)
h0 = r(2, B, H)
mask = (torch.rand(S1, B, H, generator=g) < 0.5).to(torch.uint8).to(dev)
dl = r(S1, B, K)
gr = dict(d_hidden0=f32(2, B, H), d_emb=f32(K, H), d_w_pre=f32(H, H), d_b_pre=f32(H), d_bn_w=f32(H), d_bn_b=f32(H), d_w_ih0=f32(G, H), d_w_hh0=f32(G, H),
          d_b_ih0=f32(G), d_b_hh0=f32(G), d_w_ih1=f32(G, H), d_w_hh1=f32(G, H), d_b_ih1=f32(G), d_b_hh1=f32(G), d_w_out=f32(K, H), d_b_out=f32(K))
def fwd():
    ops.code_rollout_fwd(codes, h0, None, None, wd, sv, mask, None, 0.0, 1, True, S1, B, H, K, 0)
def bwd():
    ops.code_rollout_bwd(dl, None, None, wd, sv, gr, mask, None, 0.0, S1, B, H, K, 0)
def step():
    fwd()
    if "bwd" in v:
        bwd()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
graph.replay()
torch.cuda.synchronize()
print("ok", float(sv["logits"].abs().sum()))
