"""Drop-in for `RAFTStereo.convex_upsample` (nndepth/models/raft_stereo/model.py:93-105; IGEV copy
igev_stereo/model.py:103-115; 2-channel CREStereo copy cre_stereo/model.py:110-122)."""
import torch

from . import ops


def convex_upsample(flow: torch.Tensor, mask: torch.Tensor, rate: int = 8) -> torch.Tensor:
    return ops.convex_upsample(flow.float(), mask.float(), rate)
