import sys, os
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from nndepth_amd import ops
torch.manual_seed(0)
def run(Cout,Cin,KH,KW,B,H,W,cfg=None):
    if cfg: os.environ["NND_CONV_CFG"]=cfg
    else: os.environ.pop("NND_CONV_CFG",None)
    w=torch.randn(Cout,Cin,KH,KW)/ (Cin*KH*KW)**0.5; b=torch.randn(Cout)
    x=torch.randn(B,Cin,H,W)
    conv=ops.Conv2d(w,b)
    y=conv(x.cuda()).cpu()
    ref=F.conv2d(x,w,b,padding=(KH//2,KW//2))
    err=(y-ref).abs()
    msg=""
    if err.max()>1e-3:
        for lo in range(0,Cin,8):
            x2=x.clone(); x2[:,lo:lo+8]=0
            r2=F.conv2d(x2,w,b,padding=(KH//2,KW//2))
            if (y-r2).abs().max() < err.max()*0.9: msg+=f" [drop {lo}:{lo+8} -> {(y-r2).abs().max().item():.2e}]"
    print(f"Cout={Cout} Cin={Cin} k={KH}x{KW} B={B} {H}x{W} cfg={cfg}: max err {err.max().item():.3e}{msg}", flush=True)
for cin in (72,80,96,104,128):
    run(32,cin,3,3,1,12,20,"3,1")
run(32,128,1,5,1,12,20,"3,1"); run(32,128,1,1,1,12,20,"3,1"); run(32,512,1,1,1,12,20,"3,1"); run(32,128,3,3,1,4,8,"3,1")
