"""Sweep (P, ks, wco) per update-block conv at the benchmark size; prints us per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.blocks import BasicUpdateBlock
from nndepth_amd._lib import NndError
B, H, W = 1, 68, 120
if len(sys.argv) > 3: B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8)
weightgen.fill_module_(ub, "update_block.")
ub = ub.to("cuda:0")
eng = ub.sync_engine("cuda:0")
ws = eng.workspace(B, H, W, "cuda:0"); ws.normal_()
names = eng.conv_names()
cfgs = [(p, k, w) for p in (1, 2) for k in (1, 2, 4) for w in (1, 2, 3, 4, 6, 8) if k * w <= (12 if p == 1 else 8)]
for i, nm in enumerate(names):
    res = []
    for (p, k, w) in cfgs:
        os.environ["NND_CONV_CFG"] = f"{p},{k},{w}"
        try:
            ms, fl = eng.profile_conv(i, B, H, W, 10, "cuda:0")
            res.append((ms * 1e3, p, k, w))
        except NndError:
            pass
    os.environ.pop("NND_CONV_CFG")
    ms, fl = eng.profile_conv(i, B, H, W, 10, "cuda:0")
    res.sort()
    print(f"{nm:26s} auto {ms*1e3:6.1f} us | " + "  ".join(f"P{p}k{k}w{w}:{t:.1f}" for t, p, k, w in res[:6]), flush=True)
