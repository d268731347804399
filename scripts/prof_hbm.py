"""HBM-bound kernels of the hot path at their BASELINE.json config shapes: time per call (events on the launch stream =
torch's current stream) and achieved GB/s over the ALGORITHMIC bytes (nndepth_amd/profiling.py; bench.py reports the same
rows live as roofline.hbm_group).      python scripts/prof_hbm.py [igev|raft]     (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nndepth_amd import profiling  # noqa: E402

if __name__ == "__main__":
    torch.manual_seed(0)
    dev = "cuda:0"
    if len(sys.argv) > 1 and sys.argv[1] == "igev":  # the IGEV group alone (scripts/pmc_hbm.sh)
        rows = profiling.igev_rows(dev)
    elif len(sys.argv) > 1 and sys.argv[1] == "raft":  # the RAFT-Stereo group alone
        rows = profiling.raft_rows(dev)
    else:
        rows = profiling.raft_rows(dev) + profiling.raft_rows(dev, B=8, H=48, W=156) + profiling.igev_rows(dev) + profiling.cre_rows(dev)
    print(profiling.format_rows(rows))
