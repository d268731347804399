"""Drop-in for the reference's `CorrBlock1D` (nndepth/models/raft_stereo/cost_volume.py:7-61).

Same call shape — `CorrBlock1D(fmap1, fmap2, num_levels, radius)` builds the pyramid,
`corr(coords)` samples it — so `model.corr_fn = CorrBlock1D` swaps it into an unmodified
reference model instance.  Both steps are HIP kernels (csrc/corr1d.hip).
"""
from typing import List

import torch

from . import ops


class CorrBlock1D:
    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4):
        self.num_levels = num_levels
        self.radius = radius
        self.shape = tuple(fmap1.shape)
        self._pyr = ops.corr1d_build(fmap1.float(), fmap2.float(), num_levels)

    @property
    def corr_pyramid(self) -> List[torch.Tensor]:
        """num_levels+1 views shaped (B*H*W, 1, W_l) like the reference attribute."""
        B, _, H, W = self.shape
        offs, widths, _ = ops.pyramid_layout(B, H, W, self.num_levels)
        return [self._pyr[o:o + B * H * W * w].view(B * H * W, 1, w) for o, w in zip(offs, widths)]

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        return ops.corr1d_lookup(self._pyr, coords.float(), self.num_levels, self.radius)

    @staticmethod
    def corr(fmap1: torch.Tensor, fmap2: torch.Tensor) -> torch.Tensor:
        """Level 0 only, shaped (B, H, W1, W2) like the reference static method."""
        B, _, H, W = fmap1.shape
        pyr = ops.corr1d_build(fmap1.float(), fmap2.float(), 1)
        return pyr[:B * H * W * W].view(B, H, W, W)
