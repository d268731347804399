"""Phase stamps of the k-th conv_split launch of ONE RAFT-Stereo encoder forward at 544x960 (2 frames), tile-major path as in
production.  Needs the STAMPS build: scripts/build_ablate.sh "STAMPS:-DNND_DBG_STAMPS"
    for k in 0 2 4 6; do NND_DBG_STAMP_LAUNCH=$k NND_LIB=scripts/ablate/lib_STAMPS.so python scripts/stamps_encoder.py; done
(launch order: layer1.0.conv1, layer1.0.conv2, layer1.1.conv1, layer1.1.conv2, layer2.0.conv1, layer2.0.downsample, ...)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nndepth_amd import weightgen  # noqa: E402
from nndepth_amd._lib import LIB_PATH  # noqa: E402
from nndepth_amd.raft_stereo import BaseRAFTStereo  # noqa: E402

m = BaseRAFTStereo(iters=1, context_dim=64, arithmetic="fp16x2")
m.auto_calibrate = False
weightgen.fill_module_(m)
m = m.to("cuda:0").eval()
f1, f2 = (x.to("cuda:0") for x in weightgen.synthetic_frames(100, 1, 544, 960))
with torch.no_grad():
    m.forward_fnet(f1, f2)
torch.cuda.synchronize()
raw = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * (4096 * 8))()
assert raw.nnd_debug_read_split_stamps_ns2(buf, 4096 * 8) == 0
full = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
keep = (full[:, 0] > 0) & (full[:, 4] >= full[:, 0])
a = full[keep][:, :5]
clk = full[keep]
ghz = np.median((clk[:, 6] - clk[:, 5]) / np.maximum(clk[:, 2] - clk[:, 1], 1) * 0.1)
t = (a - a[:, 0].min()) / 100.0
ph = np.diff(t, axis=1)
print(f"launch {os.environ.get('NND_DBG_STAMP_LAUNCH')}: first {len(a)} WGs: start spread {t[:, 0].max():.1f} us | prologue {ph[:, 0].mean():.1f} K-loop {ph[:, 1].mean():.1f} "
      f"reduce {ph[:, 2].mean():.1f} epilogue {ph[:, 3].mean():.1f} (per WG, us) | WG lifetime {(t[:, 4] - t[:, 0]).mean():.1f} | last end {t[:, 4].max():.1f} | K-loop clock {ghz:.2f} GHz")
