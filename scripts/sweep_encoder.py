"""Sweep (P, ks, wco) for the encoder's conv shapes at 544x960, 2 frames; prints us per launch and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import ops
from nndepth_amd._lib import NndError
DEV = "cuda:0"
shapes = [  # Cout, Cin, K, stride, Hin, Win
    (64, 64, 3, 1, 272, 480), (64, 64, 1, 1, 272, 480), (96, 64, 3, 2, 272, 480), (96, 64, 1, 2, 272, 480),
    (96, 96, 3, 1, 136, 240), (96, 96, 1, 1, 136, 240), (128, 96, 3, 2, 136, 240), (128, 128, 3, 1, 68, 120)]
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for si, (Cout, Cin, K, st, H, W) in enumerate(shapes):
    if only >= 0 and si != only:
        continue
    w = torch.randn(Cout, Cin, K, K) * 0.05
    bn = (torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout), torch.ones(Cout))
    conv = ops.ConvNorm(w, torch.zeros(Cout), st, bn, 1e-5, DEV)
    x = torch.randn(2, Cin, H, W, device=DEV)
    gf = 2.0 * 2 * ((H + st - 1) // st) * ((W + st - 1) // st) * Cout * Cin * K * K / 1e9
    res = []
    for p in (1, 2):
        for k in (1, 2):
            for wc in (1, 2, 3, 4, 6, 8):
                os.environ["NND_CONV_CFG"] = f"{p},{k},{wc}"
                try:
                    res.append((timeit(lambda: conv(x, relu=True)), p, k, wc))
                except NndError:
                    pass
    os.environ.pop("NND_CONV_CFG")
    auto = timeit(lambda: conv(x, relu=True))
    res.sort()
    print(f"{Cout:3d}<-{Cin:3d} {K}x{K} s{st} @{H}x{W}: {gf:6.2f} GF auto {auto:7.1f} us ({gf / auto * 1e3:5.1f} TF) | "
          + "  ".join(f"P{p}k{k}w{wc}:{t:.0f}" for t, p, k, wc in res[:6]), flush=True)
