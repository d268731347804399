"""Uninitialised-read detector: the same RAFT-Stereo forward with every torch.empty() of the package (outputs, workspaces) replaced
by a NaN-filled / 1e30-filled / zero-filled tensor.  A kernel that reads memory it (or an earlier kernel of the forward) did not
write shows up as a changed or non-finite output.   python scripts/poison_empty.py [arithmetic]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen, ops
from nndepth_amd.raft_stereo import BaseRAFTStereo
DEV = "cuda:0"
ar = sys.argv[1] if len(sys.argv) > 1 else "fp16x2"
real_empty, real_zeros = torch.empty, torch.zeros
fill = [None]


def poisoned_empty(*a, **k):
    t = real_empty(*a, **k)
    if fill[0] is not None and t.is_floating_point():
        t.fill_(fill[0])
    return t


def poisoned_zeros(*a, **k):  # the update-block workspace: zero only where the library says it must be
    return real_zeros(*a, **k)


def run(f):
    fill[0] = f
    torch.empty = poisoned_empty
    try:
        m = BaseRAFTStereo(iters=6, context_dim=64, arithmetic=ar)
        weightgen.fill_module_(m)
        m = m.to(DEV).eval()
        fr = tuple(x.to(DEV) for x in weightgen.synthetic_frames(20, 1, 96, 160))
        outs = [m(*fr) for _ in range(2)]
        return [o["up_disp"].clone() for o in outs[-1]]
    finally:
        torch.empty = real_empty


base = run(None)
for f in (0.0, float("nan"), 1e30, -3.0):
    got = run(f)
    bad = [(k, (got[k] - base[k]).abs().max().item(), bool(torch.isfinite(got[k]).all())) for k in range(6) if not torch.equal(got[k], base[k])]
    print(f"[{ar}] torch.empty filled with {f}: {'identical' if not bad else bad}")
