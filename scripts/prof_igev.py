"""IGEV-Stereo forward (tiny test backbone) x3 at 544x960 for rocprofv3 --kernel-trace [--stats] (trace_last.py, kernel_stats_top.py).
    python scripts/prof_igev.py [fp32|bf16x3] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from igev_double import make_igev
from nndepth_amd import weightgen
from nndepth_amd.igev_stereo import IGEVStereoBase, CostVolumeFilterNetwork
dev = "cuda:0"
ar = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = make_igev(IGEVStereoBase, CostVolumeFilterNetwork, iters=32, hidden_dim=64, context_dim=64, arithmetic=ar)
weightgen.fill_module_(m, "igev.")
m = m.to(dev).eval()
f1, f2 = weightgen.synthetic_frames(6, B, 544, 960)
f1, f2 = f1.to(dev), f2.to(dev)
for _ in range(3):
    out = m(f1, f2)
    torch.cuda.synchronize()
