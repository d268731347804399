"""Launch each update-block conv a few times at the benchmark size (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.blocks import BasicUpdateBlock
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
arith = sys.argv[2] if len(sys.argv) > 2 else "fp32"
ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8, arithmetic=arith)
weightgen.fill_module_(ub, "update_block.")
ub = ub.to("cuda:0")
eng = ub.sync_engine("cuda:0")
ws = eng.workspace(1, 68, 120, "cuda:0")
ws.normal_()
for i, nm in enumerate(eng.conv_names()):
    ms, fl = eng.profile_conv(i, 1, 68, 120, reps, "cuda:0")
    print(f"{i:2d} {nm:26s} {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF")

import ctypes as C
from nndepth_amd._lib import lib, check
scr = torch.zeros(16, device="cuda:0")
for w in (1, 2, 4):
    tf = C.c_float()
    check(lib.nnd_profile_mfma_peak(w, 20000, None, C.c_void_p(scr.data_ptr()), C.byref(tf)))
    print(f"mfma peak probe, {w} waves/SIMD: {tf.value:.1f} TFLOP/s")
