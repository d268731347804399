import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nndepth_amd import ops
torch.manual_seed(0)
for arith in ("fp16x2", "bf16x3"):
    for (Cout, Cin, KH, KW, H, W) in [(32, 16, 3, 3, 4, 8), (32, 32, 3, 3, 4, 8), (32, 64, 3, 3, 4, 8), (192, 256, 3, 3, 68, 120), (64, 64, 1, 5, 8, 16)]:
        w = torch.randn(Cout, Cin, KH, KW) / (Cin * KH * KW) ** 0.5
        b = torch.zeros(Cout)
        x = torch.randn(1, Cin, H, W)
        ref = torch.nn.functional.conv2d(x, w, b, padding=(KH // 2, KW // 2))
        y = ops.Conv2d(w, b, arithmetic=arith)(x.cuda()).cpu()
        bad = ~torch.isfinite(y)
        err = (y - ref).abs()
        err[bad] = 0
        print(os.environ.get("NND_LIB", "product")[-16:], arith, (Cout, Cin, KH, KW, H, W), "nan", int(bad.sum()), "wrong", int((err > 1e-3).sum()), "of", y.numel(), "max err", f"{float(err.max()):.3g}")
