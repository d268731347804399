"""Phase stamps of the fused flow-branch kernel (conv_split.hip: flow_branch_kernel; debug build made by
scripts/build_ablate.sh "STAMPS:-DNND_DBG_STAMPS", selected through NND_LIB) at 68x120: operands to LDS / convf1 on the VALU /
MFMA walk / K-slice sum + epilogue.
    NND_LIB=scripts/ablate/lib_STAMPS.so python scripts/stamps_flow_branch.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nndepth_amd import weightgen  # noqa: E402
from nndepth_amd.blocks import BasicUpdateBlock  # noqa: E402
from nndepth_amd._lib import LIB_PATH  # noqa: E402

H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (68, 120)
ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8, arithmetic="bf16x3")
weightgen.fill_module_(ub, "update_block.")
ub = ub.to("cuda:0").eval()
torch.manual_seed(0)
net, inp = torch.tanh(torch.randn(1, 128, H, W)).cuda(), torch.relu(torch.randn(1, 64, H, W)).cuda()
corr, flow = torch.randn(1, 36, H, W).cuda(), torch.randn(1, 1, H, W).cuda() * 3
for _ in range(3):
    ub(net, inp, corr, flow)
torch.cuda.synchronize()
raw = C.CDLL(LIB_PATH)
assert hasattr(raw, "nnd_debug_read_fb_stamps"), "needs the -DNND_DBG_STAMPS build (NND_LIB)"
buf = (C.c_ulonglong * (4096 * 8))()
assert raw.nnd_debug_read_fb_stamps(buf, 4096 * 8) == 0
a = np.array(buf[:], dtype=np.int64).reshape(4096, 8)[:, :5]
a = a[(a[:, 0] > 0) & (a[:, 4] >= a[:, 0])]
us = (a - a[:, 0].min()) / 100.0
ph = np.diff(us, axis=1)
print(f"flow_branch {H}x{W}: WGs {len(a)}, start spread {us[:, 0].max():.1f} us | operands to LDS {ph[:, 0].mean():.1f} | convf1 (VALU) "
      f"{ph[:, 1].mean():.1f} | MFMA walk {ph[:, 2].mean():.1f} | K-slice sum + epilogue {ph[:, 3].mean():.1f} | last end {us[:, 4].max():.1f}")
