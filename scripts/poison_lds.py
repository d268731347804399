"""Uninitialised-LDS-read detector: the RAFT-Stereo forward with every CU's LDS filled with a pattern between the launches of the
refinement loop (NND_DEBUG_LDS_POISON, csrc/update_block.hip: debug_sync).  A kernel that reads LDS it did not write shows up as
an output that depends on the pattern.   python scripts/poison_lds.py [arithmetic]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from nndepth_amd import weightgen
from nndepth_amd.raft_stereo import BaseRAFTStereo
m = BaseRAFTStereo(iters=6, context_dim=64, arithmetic=sys.argv[1])
weightgen.fill_module_(m)
m = m.to("cuda:0").eval()
fr = tuple(x.to("cuda:0") for x in weightgen.synthetic_frames(20, 1, 96, 160))
out = [o["up_disp"] for o in m(*fr)]
torch.save([o.cpu() for o in out], sys.argv[2])
''' % ROOT
ar = sys.argv[1] if len(sys.argv) > 1 else "fp16x2"
import torch
res = {}
for pat in (None, "0x00000000", "0xffffffff", "0x3c003c00", "0x7bff7bff", "0x40404040"):
    env = dict(os.environ)
    if pat:
        env["NND_DEBUG_LDS_POISON"] = pat
    f = f"/tmp/poison_{ar}_{pat}.pt"
    r = subprocess.run([sys.executable, "-c", WORKER, ar, f], env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print(f"[{ar}] pattern {pat}: worker failed: {r.stderr[-300:]}")
        continue
    res[pat] = torch.load(f)
base = res[None]
for pat, out in res.items():
    if pat is None:
        continue
    bad = [(k, float((out[k] - base[k]).abs().max()), bool(torch.isfinite(out[k]).all())) for k in range(6) if not torch.equal(out[k], base[k])]
    print(f"[{ar}] LDS pattern {pat}: {'identical to the unpoisoned run' if not bad else bad}")
