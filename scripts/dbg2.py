import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, torch.nn.functional as F
from nndepth_amd import weightgen
from nndepth_amd.blocks import BasicUpdateBlock
from oracle import torch_ref as R
g = dict(np.load("/root/repo/tests/golden/update_block.npz"))
name="raft_h128_c64"; hid,ctx,cp,fc=128,64,36,1
ub = BasicUpdateBlock(hidden_dim=hid, context_dim=ctx, cor_planes=cp, flow_channel=fc, spatial_scale=8)
P="ub."+name
sd = weightgen.fill_state_dict(R.update_block_spec(P, hid, cp, ctx, fc, 8))
ub.load_state_dict({k[len(P)+1:]: v for k,v in sd.items()})
ub = ub.to("cuda:0")
net,inp,corr,flow=(torch.from_numpy(g[f"{name}_{k}"]) for k in ("net","inp","corr","flow"))
n2,m2,d2 = ub(net.cuda(), inp.cuda(), corr.cuda(), flow.cuda())
torch.cuda.synchronize()
ws = ub.engine._ws.cpu()
B,_,H,W = net.shape; n=B*H*W
def carve():
    off=0; out={}
    for nm,C in (("c1",256),("cf",256),("f1",128),("hx",2*hid+ctx),("z",hid),("rh",hid),("fm",3*hid),("corr",cp),("mask",576),("delta",fc),("coords",1),("flow",fc)):
        out[nm]=ws[off:off+C*n].view(B,C,H,W); off+=(C*n+63)//64*64
    return out
w=carve()
cv=lambda nm,x,**k: F.conv2d(x, sd[P+"."+nm+".weight"], sd[P+"."+nm+".bias"], **k)
c1=torch.relu(cv("encoder.convc1",corr)); print("c1", (w["c1"]-c1).abs().max().item())
c2=torch.relu(cv("encoder.convc2",c1,padding=1)); print("c2", (w["cf"][:,:192]-c2).abs().max().item())
f1=torch.relu(cv("encoder.convf1",flow,padding=3)); print("f1", (w["f1"]-f1).abs().max().item())
f2=torch.relu(cv("encoder.convf2",f1,padding=1)); print("f2", (w["cf"][:,192:]-f2).abs().max().item())
mo=torch.relu(cv("encoder.conv",torch.cat([c2,f2],1),padding=1)); print("conv", (w["hx"][:,hid+ctx:hid+ctx+hid-fc]-mo).abs().max().item())
print("hx flow", (w["hx"][:,-fc:]-flow).abs().max().item(), "hx inp", (w["hx"][:,hid:hid+ctx]-inp).abs().max().item())
x=torch.cat([inp,mo,flow],1); h=net
hx=torch.cat([h,x],1)
z=torch.sigmoid(cv("gru.convz1",hx,padding=(0,2))); r=torch.sigmoid(cv("gru.convr1",hx,padding=(0,2)))
q=torch.tanh(cv("gru.convq1",torch.cat([r*h,x],1),padding=(0,2))); h1=(1-z)*h+z*q
hx=torch.cat([h1,x],1)
z2=torch.sigmoid(cv("gru.convz2",hx,padding=(2,0))); r2=torch.sigmoid(cv("gru.convr2",hx,padding=(2,0)))
print("z2", (w["z"]-z2).abs().max().item(), "rh2", (w["rh"]-r2*h1).abs().max().item())
q2=torch.tanh(cv("gru.convq2",torch.cat([r2*h1,x],1),padding=(2,0))); h2=(1-z2)*h1+z2*q2
print("h2", (w["hx"][:,:hid]-h2).abs().max().item(), "net_out", (n2.cpu()-h2).abs().max().item())
fm_f=torch.relu(cv("flow_head.conv1",h2,padding=1)); fm_m=torch.relu(cv("mask.0",h2,padding=1))
print("fm flow", (w["fm"][:,:hid]-fm_f).abs().max().item(), "fm mask", (w["fm"][:,hid:]-fm_m).abs().max().item())
d=cv("flow_head.conv2",fm_f,padding=1); print("delta", (d2.cpu()-d).abs().max().item())
m=0.25*cv("mask.2",fm_m); print("mask", (m2.cpu()-m).abs().max().item())
