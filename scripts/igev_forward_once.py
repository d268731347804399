"""Three IGEV forwards at 544x960 batch 1 (config 3, test backbone, fp16x2) for rocprofv3 --kernel-trace (scripts/prof_forward_kernels.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from igev_double import make_igev  # noqa: E402
from nndepth_amd import weightgen  # noqa: E402
from nndepth_amd.igev_stereo import IGEVStereoBase, CostVolumeFilterNetwork  # noqa: E402

dev = "cuda:0"
m = make_igev(IGEVStereoBase, CostVolumeFilterNetwork, iters=32, hidden_dim=64, context_dim=64, arithmetic="fp16x2")
weightgen.fill_module_(m, "igev.")
m = m.to(dev).eval()
a, b = (x.to(dev) for x in weightgen.synthetic_frames(4, 1, 544, 960))
for _ in range(3):
    m(a, b)
torch.cuda.synchronize()
