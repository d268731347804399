import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from nndepth_amd import weightgen
from nndepth_amd.raft_stereo import BaseRAFTStereo
dev = "cuda:0"
m = BaseRAFTStereo(iters=4, context_dim=64)
weightgen.fill_module_(m)
m = m.to(dev).eval()
f1, f2 = weightgen.synthetic_frames(2, 8, 384, 1248)
f1, f2 = f1.to(dev), f2.to(dev)
for _ in range(2):
    m(f1, f2)
torch.cuda.synchronize()
