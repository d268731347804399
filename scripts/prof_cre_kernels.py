"""One CREStereo 1080x1920 / 20-iteration cascade forward (random weights) for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o c -- python scripts/prof_cre_kernels.py [fp32|bf16x3]
then scripts/kernel_stats_top.py OUT prints the per-kernel totals of the LAST forward's share."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nndepth_amd import weightgen  # noqa: E402
from nndepth_amd.cre_stereo import CREStereoBase  # noqa: E402

ar = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
m = CREStereoBase(iters=20, arithmetic=ar).to(dev).eval()
weightgen.fill_module_(m)

l, r = (x.to(dev) for x in weightgen.synthetic_frames(3, 1, 1080, 1920))

with torch.no_grad():
    for _ in range(reps):
        out = m(l, r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = m(l, r)
    e1.record()
    torch.cuda.synchronize()
print(f"CREStereo 1080x1920 20 iters [{ar}]: {e0.elapsed_time(e1):.2f} ms")
