"""Phase timeline inside conv_mfma (debug build with -DNND_DBG_STAMPS, NND_LIB=scripts/libstamps.so):
per workgroup s_memrealtime stamps 0 entry, 1 prologue done, 2 K-loop done, 3 reduction done, 4 end."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nndepth_amd import weightgen
from nndepth_amd.blocks import BasicUpdateBlock
from nndepth_amd._lib import lib, LIB_PATH
raw = C.CDLL(LIB_PATH)
ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8)
weightgen.fill_module_(ub, "update_block.")
ub = ub.to("cuda:0"); eng = ub.sync_engine("cuda:0")
ws = eng.workspace(1, 68, 120, "cuda:0"); ws.normal_()
names = eng.conv_names()
buf = (C.c_ulonglong * (4096 * 8))()
for i, nm in enumerate(names):
    eng.profile_conv(i, 1, 68, 120, 1, "cuda:0")   # warm + 1 rep: stamps hold the last launch
    torch.cuda.synchronize()
    assert raw.nnd_debug_read_stamps(buf, 4096 * 8) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(4096, 8)[:, :5]
    a = a[(a[:, 0] > 0) & (a[:, 4] >= a[:, 0])]
    t0 = a[:, 0].min()
    us = (a - t0) / 100.0   # 100 MHz -> us
    ph = np.diff(us, axis=1)
    print(f"{nm:30s} WGs {len(a):4d}  start spread {us[:,0].max():5.1f} us | prologue {ph[:,0].mean():5.1f}  K-loop {ph[:,1].mean():6.1f}  reduce {ph[:,2].mean():4.1f}  epilogue {ph[:,3].mean():5.1f} | last end {us[:,4].max():6.1f} us, mean WG life {(us[:,4]-us[:,0]).mean():6.1f}")
