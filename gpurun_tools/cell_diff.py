import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gesture2vec_amd import ops
DEV="cuda:0"
B,IN,H=24,48,48
g=torch.Generator().manual_seed(17)
x,h=torch.randn(B,IN,generator=g).to(DEV),(torch.randn(B,H,generator=g)*0.5).to(DEV)
wi,wh=(torch.randn(3*H,IN,generator=g)*0.1).to(DEV),(torch.randn(3*H,H,generator=g)*0.1).to(DEV)
bi,bh=(torch.randn(3*H,generator=g)*0.1).to(DEV),(torch.randn(3*H,generator=g)*0.1).to(DEV)
gates=torch.empty(B,4*H,device=DEV)
hn=ops.gru_cell_fwd(x,h,wi,wh,bi,bh,gates=gates)
gi=ops.linear_fwd(x,wi,bi)
hs2,hn2,g2=ops.gru_seq_fwd(gi.view(1,B,3*H),wh,bh,1,B,H,h0=h)
g2=g2.view(B,4*H)
print("h", float((hn-hn2).abs().max()))
for k,n in enumerate("r z n gh".split()):
    print(n, float((gates[:,k*H:(k+1)*H]-g2[:,k*H:(k+1)*H]).abs().max()))
ref_gh = h @ wh[2*H:].t() + bh[2*H:]
print("gh vs torch", float((gates[:,3*H:]-ref_gh).abs().max()), float((g2[:,3*H:]-ref_gh).abs().max()))
