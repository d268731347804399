"""IGEV-Stereo forward at 544x960 on the tiny test backbone of tests/igev_double.py (the real backbone is timm's
MobileNetV3): shows where the time goes between the PyTorch regulariser (a15) and the HIP hot path (a12-a14, a16)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from igev_double import make_igev
from nndepth_amd import weightgen
from nndepth_amd.igev_stereo import IGEVStereoBase, CostVolumeFilterNetwork
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = make_igev(IGEVStereoBase, CostVolumeFilterNetwork, iters=32, hidden_dim=64, context_dim=64)
weightgen.fill_module_(m, "igev.")
m = m.to(dev).eval()
f1, f2 = weightgen.synthetic_frames(6, B, 544, 960)
f1, f2 = f1.to(dev), f2.to(dev)
for _ in range(2):
    out = m(f1, f2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = m(f1, f2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"IGEV 544x960 batch {B}, 32 iters (tiny backbone): {dt * 1e3:.1f} ms / batch = {B / dt:.2f} pairs/s; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
