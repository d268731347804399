"""Time of IGEV's cv_squeezer Conv3d(8->1) + soft-argmin on PyTorch-ROCm vs the fused HIP kernel (544x960: 136x240x240)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import ops
dev = "cuda:0"
for B in (1, 4):
    G, H, W = 8, 136, 240
    geo = torch.randn(B, G, H, W, W, device=dev)
    conv = torch.nn.Conv3d(G, 1, 3, 1, 1).to(dev)
    def torch_path():
        logits = conv(geo.permute(0, 1, 4, 2, 3)).squeeze(1)
        return ops.softargmin_disparity(logits.float())
    with torch.no_grad():
        for _ in range(2): ref = torch_path()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): torch_path()
        torch.cuda.synchronize(); t_t = (time.perf_counter() - t0) / 5
        line = f"B={B}: PyTorch Conv3d + softargmin {t_t * 1e3:.2f} ms"
        if hasattr(ops, "igev_init_disparity"):
            for _ in range(2): out = ops.igev_init_disparity(geo, conv.weight, conv.bias, B, G, H, W, W)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): ops.igev_init_disparity(geo, conv.weight, conv.bias, B, G, H, W, W)
            torch.cuda.synchronize(); t_h = (time.perf_counter() - t0) / 5
            line += f", fused HIP {t_h * 1e3:.2f} ms, max-abs {float((out - ref).abs().max()):.2e} (|init| <= {float(ref.abs().max()):.1f})"
    print(line)
    del geo
