import sys; sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from nndepth_amd import weightgen, ops
from nndepth_amd.raft_stereo import BaseRAFTStereo
from nndepth_amd.cost_volume import CorrBlock1D
from oracle import torch_ref as R
sd = weightgen.fill_state_dict(R.raft_stereo_spec())
m = BaseRAFTStereo(iters=1, context_dim=64); m.load_state_dict(sd); m = m.cuda().eval()
f1, f2 = weightgen.synthetic_frames(0, 1, 96, 160)
with torch.no_grad():
    out = m(f1.cuda(), f2.cuda())
    torch.cuda.synchronize()
    eng = m.update_block.engine
    ws = eng._ws.cpu()
    B,H,W = 1,12,20; n=B*H*W; hid,ctx,cp,fc=128,64,36,1
    off=0; bufs={}
    for nm,C in (("c1",256),("cf",256),("f1",128),("hx",2*hid+ctx),("z",hid),("rh",hid),("fm",3*hid),("corr",cp),("mask",576),("delta",fc),("coords",1),("flow",fc),("hcopy",2*hid)):
        bufs[nm]=ws[off:off+C*n].view(B,C,H,W); off+=(C*n+63)//64*64
    x = bufs["fm"][:, hid:]
    flow = bufs["flow"]
    wt, bs = sd["update_block.mask.2.weight"], sd["update_block.mask.2.bias"]
    ref = R.convex_upsample(flow, 0.25*F.conv2d(x, wt, bs), 8)
    got = out[0]["up_disp"].cpu()
    print("fused-in-loop vs torch from workspace:", (got-ref).abs().max().item())
    # mask.0 check
    h = bufs["hx"][:, :hid]
    m0 = torch.relu(F.conv2d(h, sd["update_block.mask.0.weight"], sd["update_block.mask.0.bias"], padding=1))
    print("mask.0 in workspace vs torch(h in ws):", (x-m0).abs().max().item(), " hcopy0 vs h:", (bufs["hcopy"][:, :hid]-h).abs().max().item())
    conv = ops.Conv2d(wt, bs)
    iso = ops.mask_upsample(conv, x.cuda(), flow.cuda(), 8).cpu()
    print("isolated fused on ws inputs vs torch:", (iso-ref).abs().max().item())
    err=(got-ref).abs()[0,0]
    bad=(err>1e-5).nonzero()
    print("nbad",len(bad),"of",err.numel())
    print("bad lowres rows", sorted(set((bad[:,0]//8).tolist())), "cols", sorted(set((bad[:,1]//8).tolist())))
    print("sub i", sorted(set((bad[:,0]%8).tolist())), "sub j", sorted(set((bad[:,1]%8).tolist())))
    # is it consistent with using the OLD flow (before advance)? flow_old = flow - delta
    flow_old = flow - bufs["delta"]
    ref_old = R.convex_upsample(flow_old, 0.25*F.conv2d(x, wt, bs), 8)
    print("vs ref with pre-advance flow:", (got-ref_old).abs().max().item())
