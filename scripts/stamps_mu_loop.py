"""Phase stamps of the fused mask.2 + softmax + upsample kernel as the RAFT-Stereo loop runs it (split arithmetic, c4 tile-major x):
the last launch of a short forward.  Needs mask_upsample.hip built with -DNND_DBG_STAMPS (scripts/ablate/lib_MUSTAMPS.so)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nndepth_amd import weightgen  # noqa: E402
from nndepth_amd._lib import LIB_PATH  # noqa: E402
from nndepth_amd.raft_stereo import BaseRAFTStereo  # noqa: E402

m = BaseRAFTStereo(iters=4, context_dim=64, arithmetic=sys.argv[1] if len(sys.argv) > 1 else "fp16x2")
weightgen.fill_module_(m)
m = m.to("cuda:0").eval()
f1, f2 = (x.to("cuda:0") for x in weightgen.synthetic_frames(100, 1, 544, 960))
for _ in range(3):
    m(f1, f2)
torch.cuda.synchronize()
raw = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * (4096 * 8))()
assert raw.nnd_debug_read_mu_stamps(buf, 4096 * 8) == 0
a = np.array(buf[:], dtype=np.int64).reshape(4096, 8)[:, :5]
a = a[(a[:, 0] > 0) & (a[:, 4] >= a[:, 0])]
us = (a - a[:, 0].min()) / 100.0
ph = np.diff(us, axis=1)
print(f"mask.2 + upsample in the loop: WGs {len(a)}: start spread {us[:, 0].max():.1f} us | stage {ph[:, 0].mean():.1f} | K loop {ph[:, 1].mean():.1f} | "
      f"K-half sum {ph[:, 2].mean():.1f} | softmax + store {ph[:, 3].mean():.1f} | last end {us[:, 4].max():.1f}")
