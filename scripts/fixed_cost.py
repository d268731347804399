"""Fixed per-launch cost of conv_mfma: time tiny convs (1 K-chunk) back to back with hipEvents."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import ops
torch.manual_seed(0)
H, W = 68, 120
def t(conv, x, reps=50):
    y = conv(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): conv(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (Cout, Cin, KH, KW) in [(128, 32, 1, 1), (128, 32, 3, 3), (256, 32, 3, 3), (128, 128, 3, 3), (128, 256, 3, 3), (128, 320, 1, 5)]:
    conv = ops.Conv2d(torch.randn(Cout, Cin, KH, KW) * 0.05, torch.randn(Cout))
    x = torch.randn(1, Cin, H, W, device="cuda")
    us = t(conv, x)
    mfma_us = 2.0 * Cout * Cin * KH * KW * H * W / 154.5e12 * 1e6
    print(f"conv {KH}x{KW} {Cin:3d}->{Cout:3d}: {us:6.1f} us per launch (incl. python/ctypes ~?), pure-MFMA time at peak {mfma_us:5.1f} us")
# empty-ish torch kernel for the python+launch floor
x = torch.zeros(1024, device="cuda")
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): x.add_(1.0)
e1.record(); torch.cuda.synchronize(); print(f"torch add_ floor: {e0.elapsed_time(e1)/200*1e3:.1f} us")
