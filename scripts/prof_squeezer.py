"""igev_init_disparity at the IGEV config shape: row-walking kernel vs the one-row kernel (NND_IGEV_SQUEEZE_V1), graph-timed.
     python scripts/prof_squeezer.py     (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nndepth_amd import _lib, ops, profiling  # noqa: E402

if __name__ == "__main__":
    torch.manual_seed(0)
    dev = "cuda:0"
    for (B, G, H, W) in ((1, 8, 136, 240), (1, 8, 68, 120), (1, 8, 96, 312)):
        geo = torch.randn(B * G * H * W, W, device=dev)
        conv = torch.nn.Conv3d(G, 1, 3, 1, 1)
        res = {}
        for name, env in (("walk", None), ("one-row", "NND_IGEV_SQUEEZE_V1")):
            if env:
                os.environ[env] = "1"
            _lib.lib.nnd_reload_switches()
            out = ops.igev_init_disparity(geo, conv.weight, conv.bias, B, G, H, W, W)
            us = profiling.time_us(lambda: ops.igev_init_disparity(geo, conv.weight, conv.bias, B, G, H, W, W), 12)
            if env:
                del os.environ[env]
                _lib.lib.nnd_reload_switches()
            res[name] = (us, out)
        d = (res["walk"][1] - res["one-row"][1]).abs().max().item()
        mb = (B * G * H * W * W + B * H * W) * 4 / 1e6
        print(f"{B}x{G}x{H}x{W}x{W}: walk {res['walk'][0]:7.2f} us ({mb / res['walk'][0] / 8e3 * 1e3:.3f} of 8 TB/s)   "
              f"one-row {res['one-row'][0]:7.2f} us   max |diff| {d:.2e}", flush=True)
