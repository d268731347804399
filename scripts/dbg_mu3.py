import sys; sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from nndepth_amd import weightgen, ops
from nndepth_amd.raft_stereo import BaseRAFTStereo
from oracle import torch_ref as R
sd = weightgen.fill_state_dict(R.raft_stereo_spec())
f1, f2 = weightgen.synthetic_frames(0, 1, 96, 160)
ref = R.raft_stereo_forward(sd, f1, f2, 2)
m = BaseRAFTStereo(iters=2, context_dim=64); m.load_state_dict(sd); m = m.cuda().eval()
for trial in range(3):
    out = m(f1.cuda(), f2.cuda())
    print("trial", trial, [ (out[i]["up_disp"].cpu()-ref[i]).abs().max().item() for i in range(2)])
m2 = BaseRAFTStereo(iters=2, context_dim=64, fused_loop=False); m2.load_state_dict(sd); m2 = m2.cuda().eval()
out = m2(f1.cuda(), f2.cuda())
print("seam path", [ (out[i]["up_disp"].cpu()-ref[i]).abs().max().item() for i in range(2)])
