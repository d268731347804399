"""Phase stamps of the fused mask.2 + softmax + convex-upsample kernel (debug build: scripts/ablate/lib_MUSTAMPS.so made with
-DNND_DBG_STAMPS, selected through NND_LIB) at 68x120, rate 8, 256 input channels: stage / K loop / K-half sum / softmax+store.
    NND_LIB=scripts/ablate/lib_MUSTAMPS.so python scripts/stamps_mu.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nndepth_amd import ops  # noqa: E402
from nndepth_amd._lib import LIB_PATH  # noqa: E402

H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (68, 120)
torch.manual_seed(0)
conv = ops.Conv2d(torch.randn(576, 256, 1, 1) * 0.05, torch.randn(576) * 0.1, "cuda:0")
x, flow = torch.relu(torch.randn(1, 256, H, W, device="cuda:0")), torch.randn(1, 1, H, W, device="cuda:0")
for _ in range(3):
    out = ops.mask_upsample(conv, x, flow, 8)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = ops.mask_upsample(conv, x, flow, 8)
e1.record()
torch.cuda.synchronize()
print(f"mask_upsample {H}x{W}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (NCHW inputs)")
raw = C.CDLL(LIB_PATH)
if hasattr(raw, "nnd_debug_read_mu_stamps"):
    buf = (C.c_ulonglong * (4096 * 8))()
    out = ops.mask_upsample(conv, x, flow, 8)
    torch.cuda.synchronize()
    assert raw.nnd_debug_read_mu_stamps(buf, 4096 * 8) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(4096, 8)[:, :5]
    a = a[(a[:, 0] > 0) & (a[:, 4] >= a[:, 0])]
    us = (a - a[:, 0].min()) / 100.0
    ph = np.diff(us, axis=1)
    print(f"WGs {len(a)}: start spread {us[:, 0].max():.1f} us | stage {ph[:, 0].mean():.1f} | K loop {ph[:, 1].mean():.1f} | "
          f"K-half sum {ph[:, 2].mean():.1f} | softmax + store {ph[:, 3].mean():.1f} | last end {us[:, 4].max():.1f}")
