"""CREStereo 1080x1920 / 20 iterations single-call forward x3 for rocprofv3 --kernel-trace (see trace_last.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.cre_stereo import CREStereoBase
dev = "cuda:0"
m = CREStereoBase(iters=20)
weightgen.fill_module_(m)
m = m.to(dev).eval()
f1, f2 = weightgen.synthetic_frames(3, 1, 1080, 1920)
f1, f2 = f1.to(dev), f2.to(dev)
for _ in range(3):
    out = m(f1, f2)
    torch.cuda.synchronize()
