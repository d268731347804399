"""One thin stride-1 Conv3d layer of the IGEV regulariser (csrc/slab3d.hip with fp16x2) a few times, for rocprofv3 --pmc /
--kernel-trace:   python scripts/prof_slab3d.py Cin Cout D H W [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import ops
Cin, Cout, D, H, W = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
torch.manual_seed(0)
w = torch.randn(Cout, Cin, 3, 3, 3) * (2.0 / (Cin * 27)) ** 0.5
conv = ops.Conv3dNorm(w, None, 1, None, 1e-5, 0.01, 0, "cuda:0", arithmetic="fp16x2")
x = torch.randn(1, D + 2, Cin, H, W, device="cuda:0")
x[:, 0] = 0
x[:, -1] = 0
for _ in range(reps):
    y = conv(x)
torch.cuda.synchronize()
print("ok", float(y.abs().max()))
