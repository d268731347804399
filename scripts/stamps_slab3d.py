"""Phase timeline of one workgroup of the depth-marching Conv3d kernel (csrc/slab3d.hip built with -DNND_SLAB3D_STAMPS, selected
through NND_LIB): per depth step the cycles from step start to [MFMA walk issued | barrier passed | next slices split + written to
LDS | loads of the step after issued | epilogue stores issued] and the step total (incl. the closing barrier).   python scripts/stamps_slab3d.py Cin Cout D H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nndepth_amd import ops
Cin, Cout, D, H, W = (int(x) for x in sys.argv[1:6])
torch.manual_seed(0)
w = torch.randn(Cout, Cin, 3, 3, 3) * (2.0 / (Cin * 27)) ** 0.5
conv = ops.Conv3dNorm(w, None, 1, None, 1e-5, 0.01, 0, "cuda:0", arithmetic="fp16x2")
x = torch.randn(1, D + 2, Cin, H, W, device="cuda:0")
for _ in range(3):
    y = conv(x)
torch.cuda.synchronize()
st = y[0, 0].reshape(-1)[:4096].view(torch.int64).cpu().numpy()
od = 2 if Cout == 8 else 1
per = 6
n = (np.count_nonzero(st) - 1) // per
a = st[:n * per].reshape(n, per)
end = np.append(a[1:, 0], st[n * per])
rel = a - a[:, :1]
names = ["MFMA walk issued", "barrier (Cout 8)", "slices split + written", "next loads issued", "affine + stores issued"]
print(f"conv3d {Cin}->{Cout} {D}x{H}x{W}: {n} steps of workgroup 9; cycles (100 MHz s_memtime ticks are NOT used: clock64 = shader clock)")
for k, nm in enumerate(names):
    print(f"  {nm:52s} median {np.median(rel[1:-1, k + 1]):8.0f}")
print(f"  {'step total':52s} median {np.median((end - a[:, 0])[1:-1]):8.0f}")
