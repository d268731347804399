"""IGEV cost-volume regulariser at the 544x960 shape (1 x 8 x 240 x 136 x 240): HIP path vs the module's PyTorch ops."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.igev_stereo import CostVolumeFilterNetwork
dev = "cuda:0"
reg = CostVolumeFilterNetwork(8, [40, 80, 160]).to(dev).eval()
weightgen.fill_module_(reg, "igev.cv_regularizer.")
x = torch.randn(1, 8, 240, 136, 240, device=dev)
feats = [torch.rand(1, 40, 68, 120, device=dev), torch.rand(1, 80, 34, 60, device=dev), torch.rand(1, 160, 17, 30, device=dev)]
with torch.no_grad():
    outs = {}
    for hip in (True, False):
        reg.hip = hip
        for _ in range(2):
            y = reg(x, feats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = reg(x, feats)
        torch.cuda.synchronize()
        outs[hip] = y
        print(f"regulariser {'HIP' if hip else 'PyTorch-ROCm'}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per sample")
    print("max-abs HIP vs PyTorch-ROCm:", (outs[True] - outs[False]).abs().max().item())
