"""Phase stamps (debug build, NND_LIB=scripts/libstamps.so) for the encoder's conv shapes."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nndepth_amd import ops
from nndepth_amd._lib import LIB_PATH
raw = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * (4096 * 8))()
for (Cout, Cin, K, st, H, W) in [(64, 64, 3, 1, 272, 480), (96, 96, 3, 1, 136, 240), (128, 128, 3, 1, 68, 120), (64, 64, 1, 1, 272, 480)]:
    w = torch.randn(Cout, Cin, K, K) * 0.05
    bn = (torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout), torch.ones(Cout))
    conv = ops.ConvNorm(w, torch.zeros(Cout), st, bn, 1e-5, "cuda:0")
    x = torch.randn(2, Cin, H, W, device="cuda:0")
    for _ in range(2):
        conv(x, relu=True)
    torch.cuda.synchronize()
    assert raw.nnd_debug_read_stamps(buf, 4096 * 8) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(4096, 8)[:, :5]
    a = a[(a[:, 0] > 0) & (a[:, 4] >= a[:, 0])]
    t0 = a[:, 0].min()
    us = (a - t0) / 100.0
    ph = np.diff(us, axis=1)
    print(f"{Cout}<-{Cin} {K}x{K} s{st} @{H}x{W}: first 4096 WGs: start spread {us[:,0].max():6.1f} us | prologue {ph[:,0].mean():5.1f}  K-loop {ph[:,1].mean():6.1f}  reduce {ph[:,2].mean():4.1f}  epilogue {ph[:,3].mean():5.1f} | mean WG life {(us[:,4]-us[:,0]).mean():6.1f}")
