"""Phase stamps of ONE stand-alone conv_split launch of an arbitrary shape through ops.Conv2d (NCHW source and destination), e.g. the
encoder's layer1 convs: 64 -> 64 3x3 at 272x480, batch 2.  Needs a -DNND_DBG_STAMPS build of the split unit:
    scripts/build_ablate.sh "STAMPS:-DNND_DBG_STAMPS";  NND_LIB=scripts/ablate/lib_STAMPS.so python scripts/stamps_conv2d.py 64 64 3 3 2 272 480 [fp16x2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nndepth_amd import ops  # noqa: E402
from nndepth_amd._lib import LIB_PATH  # noqa: E402

Cout, Cin, KH, KW, B, H, W = (int(a) for a in sys.argv[1:8])
arith = sys.argv[8] if len(sys.argv) > 8 else "fp16x2"
torch.manual_seed(0)
conv = ops.Conv2d(torch.randn(Cout, Cin, KH, KW) / (Cin * KH * KW) ** 0.5, torch.randn(Cout), arithmetic=arith)
x = torch.randn(B, Cin, H, W, device="cuda:0")
for _ in range(3):
    y = conv(x, relu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = conv(x, relu=True)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
raw = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * (4096 * 8))()
ns = {"fp16x2": 2, "bf16x3": 3}[arith]
fn = getattr(raw, f"nnd_debug_read_split_stamps_ns{ns}", None)
line = f"conv {Cin}->{Cout} {KH}x{KW} at {B}x{H}x{W} {arith}: {us:.1f} us per launch ({2.0 * B * H * W * Cout * Cin * KH * KW / us / 1e6:.0f} TFLOP/s algorithmic)"
if fn is not None and fn(buf, 4096 * 8) == 0:
    full = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
    keep = (full[:, 0] > 0) & (full[:, 4] >= full[:, 0])
    a = full[keep][:, :5]
    clk = full[keep]
    ghz = np.median((clk[:, 6] - clk[:, 5]) / np.maximum(clk[:, 2] - clk[:, 1], 1) * 0.1)
    t = (a - a[:, 0].min()) / 100.0
    ph = np.diff(t, axis=1)
    line += (f" | first {len(a)} WGs: start spread {t[:, 0].max():.1f} us | prologue {ph[:, 0].mean():.1f} K-loop {ph[:, 1].mean():.1f} reduce {ph[:, 2].mean():.1f} "
             f"epilogue {ph[:, 3].mean():.1f} (per WG, us) | WG lifetime {(t[:, 4] - t[:, 0]).mean():.1f} | K-loop clock {ghz:.2f} GHz")
print(line)
