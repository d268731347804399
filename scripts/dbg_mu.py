import sys; sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from nndepth_amd import ops
from oracle import torch_ref as R
torch.manual_seed(0)
for (B,Cin,H,W,r) in [(1,256,12,20,8),(1,256,8,8,8),(2,256,9,13,8),(1,128,8,16,4)]:
    w=torch.randn(9*r*r,Cin,1,1)/Cin**0.5; b=torch.randn(9*r*r); x=torch.relu(torch.randn(B,Cin,H,W)); flow=torch.randn(B,1,H,W)*5
    conv=ops.Conv2d(w,b)
    out=ops.mask_upsample(conv,x.cuda(),flow.cuda(),r).cpu()
    ref=R.convex_upsample(flow,0.25*F.conv2d(x,w,b),r)
    err=(out-ref).abs()
    print((B,Cin,H,W,r),"max err",err.max().item())
    if err.max()>1e-3:
        bad=(err>1e-3).nonzero(); print("  nbad",len(bad),"of",err.numel(),"rows",sorted(set((bad[:,2]%r).tolist())),"cols%r",sorted(set((bad[:,3]%r).tolist())),"tilecol",sorted(set((bad[:,3]//(8*r)).tolist())),"px row in tile",sorted(set(((bad[:,2]//r)%4).tolist())),"px col in tile",sorted(set(((bad[:,3]//r)%8).tolist())))
