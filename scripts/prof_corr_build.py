"""corr1d_build at the two RAFT shapes and the IGEV group shape, k-split kernel vs LDS-staged kernel (NND_CORR_BUILD_NO_KSPLIT):
graph-timed µs per launch.     python scripts/prof_corr_build.py     (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nndepth_amd import _lib, ops, profiling  # noqa: E402

if __name__ == "__main__":
    torch.manual_seed(0)
    dev = "cuda:0"
    for (B, C, H, W) in ((1, 256, 68, 120), (2, 256, 68, 120), (8, 256, 48, 156), (1, 256, 34, 60), (1, 128, 136, 240)):
        f1, f2 = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
        res = {}
        for name, env in (("ksplit", None), ("staged", "NND_CORR_BUILD_NO_KSPLIT")):
            if env:
                os.environ[env] = "1"
            _lib.lib.nnd_reload_switches()
            out = ops.corr1d_build(f1, f2, 4)
            us = profiling.time_us(lambda: ops.corr1d_build(f1, f2, 4), 50)
            if env:
                del os.environ[env]
                _lib.lib.nnd_reload_switches()
            res[name] = (us, out)
        d = (res["ksplit"][1] - res["staged"][1]).abs().max().item()
        print(f"{B}x{C}x{H}x{W}: k-split {res['ksplit'][0]:7.2f} us   staged {res['staged'][0]:7.2f} us   max |diff| {d:.2e}", flush=True)
