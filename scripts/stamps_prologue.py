"""Where the prologue of a conv_split launch goes (kernel start -> first barrier): real-time stamps of workgroup thread 0 at
kernel start | units decoded | weight ring + first patch requested | first patch split and stored | behind the barrier.
Needs scripts/ablate/lib_PROLOGUE.so (scripts/build_ablate.sh "PROLOGUE:-DNND_DBG_STAMPS -DNND_DBG_PROLOGUE").
    NND_LIB=scripts/ablate/lib_PROLOGUE.so python scripts/stamps_prologue.py     (on the GPU box)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nndepth_amd import weightgen  # noqa: E402
from nndepth_amd._lib import LIB_PATH  # noqa: E402
from nndepth_amd.blocks import BasicUpdateBlock  # noqa: E402

ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8, arithmetic="fp16x2")
weightgen.fill_module_(ub, "update_block.")
eng = ub.to("cuda:0").eval().sync_engine("cuda:0")
ws = eng.workspace(1, 68, 120, "cuda:0")
ws.normal_()
raw = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * (4096 * 8))()
for i, nm in enumerate(eng.conv_names()):
    if nm in ("encoder.convc1", "mask.2"):
        continue
    eng.profile_conv(i, 1, 68, 120, 5, "cuda:0")
    torch.cuda.synchronize()
    assert raw.nnd_debug_read_split_stamps_ns2(buf, 4096 * 8) == 0
    f = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
    f = f[(f[:, 0] > 0) & (f[:, 1] >= f[:, 0])]
    seq = f[:, [0, 5, 6, 7, 1]] / 100.0
    ph = np.diff(seq, axis=1)
    print(f"{nm:30s} WGs {len(f):4d} | decode {ph[:, 0].mean():5.2f} | issue loads {ph[:, 1].mean():5.2f} | wait + split + store {ph[:, 2].mean():5.2f} | "
          f"barrier {ph[:, 3].mean():5.2f} | prologue {(seq[:, 4] - seq[:, 0]).mean():5.2f} us", flush=True)
