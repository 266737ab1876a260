import sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
from stcat_amd import _lib as L, ops
import os
L.load(); L.set_mma_mode(os.environ.get("MODE","f16x3p")); L.call("stcat_set_f16_scales", 6, 2)
dev = torch.device("cuda:0")
torch.manual_seed(0)
n,H,W,Cin,Cout,k,stride,pad = 4,28,28,256,256,3,1,1
x = torch.randn(n,H,W,Cin,device=dev); w = torch.randn(Cout,k,k,Cin,device=dev)*(Cin*k*k)**-0.5
g = torch.randn(n,H,W,Cout,device=dev)
cache = ops.WeightPlanes(); wp, wt = cache.refresh([w], transposed=True); wt = wt[w.data_ptr()]
G,_ = ops.pl_act_bwd_raw(g, None, None, want_g=True, want_res=False, relu=False)
GS = L.f16_grad_scale()
ref = F.conv_transpose2d(g.permute(0,3,1,2).double(), w.permute(0,3,1,2).double(), stride=stride, padding=pad).permute(0,2,3,1)
for tile in (-1, 3):
    L.call("stcat_debug_force_pl_tile", tile)
    dx = ops.pl_conv_dgrad_raw(G, wt, x.shape, k, stride, pad)
    msc = torch.rand(Cin, device=dev)+0.5
    ym = ops.pl_split(torch.ones_like(x)); ym.mask = torch.full((n*H*W, Cin//8), 255, dtype=torch.uint8, device=dev)
    dx4, dx5 = ops.pl_conv_dgrad_raw(G, wt, x.shape, k, stride, pad, add=dx, mask_y=ym, scale2=msc)
    e1 = (ops.pl_join(dx).double()/GS - ref).abs().max().item()
    e4 = (ops.pl_join(dx4).double()/GS - 2*ref).abs().max().item()
    e5 = (ops.pl_join(dx5).double()/GS - 2*ref*msc.double()).abs().max().item()
    print(tile, "dx", e1, "dx4", e4, "dx5", e5, "scale", ref.abs().max().item())
    t = dx5.t.view(torch.float16).float()
    want = (2*ref*msc.double()*GS)
    print("   hi only err", (t[0].double()-want).abs().max().item(), " hi+lo err", ((t[0].double()+t[1].double())-want).abs().max().item(), " lo absmax", t[1].abs().max().item(), " want-hi absmax", (want-t[0].double()).abs().max().item())
    idx = ((t[0].double()+t[1].double())-want).abs().argmax().item()
    print("   worst idx", idx, "want", want.reshape(-1)[idx].item(), "hi", t[0].reshape(-1)[idx].item(), "lo", t[1].reshape(-1)[idx].item(), "chan", idx % Cin, "msc", msc[idx % Cin].item())
