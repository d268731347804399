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


class GroupCorrBlock1D:
    """Drop-in for `GroupCorrBlock1D` (nndepth/models/raft_stereo/cost_volume.py:64-128; the correlation of
    Coarse2FineGroupRepViTRAFTStereo): same constructor arguments, `corr_pyramid` (num_levels+1 entries shaped
    (B*G*H*W, 1, W_l)) and `__call__(coords)`.  Both reference quirks are kept: only the first `num_groups` chunks of `num_groups`
    channels are correlated, scaled by 1/sqrt(C_total) (Q4), and the lookup views its (B*G*H*W, 2r+1) samples as (B, H, W, -1)
    without moving the group axis (Q6)."""

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4, num_groups: int = 4):
        self.num_levels, self.radius, self.num_groups = num_levels, radius, num_groups
        self.shape = tuple(fmap1.shape)
        self._pyr = ops.raft_group_corr_build(fmap1.float(), fmap2.float(), num_groups, num_levels)

    @property
    def corr_pyramid(self) -> List[torch.Tensor]:
        B, _, H, W = self.shape
        rows = B * self.num_groups * H * W
        offs, widths, _ = ops.pyramid_layout(B * self.num_groups, H, W, self.num_levels)
        return [self._pyr[o:o + rows * w].view(rows, 1, w) for o, w in zip(offs, widths)]

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        return ops.group_corr1d_lookup(self._pyr, coords.float(), self.num_groups, self.num_levels, self.radius)

    def corr(self, fmap1: torch.Tensor, fmap2: torch.Tensor) -> torch.Tensor:
        """Level 0 only, shaped (B, G, H, W1, W2) like the reference method."""
        B, _, H, W = fmap1.shape
        pyr = ops.raft_group_corr_build(fmap1.float(), fmap2.float(), self.num_groups, 1)
        return pyr[:B * self.num_groups * H * W * W].view(B, self.num_groups, H, W, W)


class GeometryAwareCostVolume:
    """Drop-in for IGEV's `GeometryAwareCostVolume` (nndepth/models/igev_stereo/cost_volume.py:9-98): same
    constructor arguments, `feat_corr_cv` / `geo_aware_cv` pyramid lists (num_levels+1 entries shaped
    (B*G*H*W, 1, W_l), `forward()` reads `geo_aware_cv[0]`, igev_stereo/model.py:144) and `corr(coords)`.
    The group-wise correlation, both avg-pool pyramids and the combined lookup are HIP kernels; the 3-D
    regulariser passed in (`regularizer_3d`, Conv3d hourglass) runs on PyTorch-ROCm."""

    def __init__(self, fmap1, fmap2, features, regularizer_3d, num_levels: int = 4, radius: int = 4, num_groups: int = 8):
        self.num_groups, self.num_levels, self.radius = num_groups, num_levels, radius
        B, C, H, W = fmap1.shape
        assert C % num_groups == 0, "Number of channels of fmap1 and fmap2 must be the factor of num_groups"
        self.shape = (B, H, W)
        # torch.split(fmap, num_groups) yields chunks of num_groups channels; only the first num_groups are used (Q4)
        hip_reg = getattr(regularizer_3d, "hip_active", None)
        lazy = hip_reg is not None and ops.igev_interleave_level0_supported(num_groups, W, num_levels)
        # lazy: level 0 of both volumes now, the pooled levels when somebody reads them — the fused loop does not (it gathers
        # from interleaved(), which pools in LDS)
        self._feat_buf = ops.group_corr_build(fmap1.float(), fmap2.float(), num_groups, num_groups, num_levels, pooled=not lazy)
        self._pooled = not lazy
        feat0 = self._feat_buf[:B * num_groups * H * W * W].view(B, num_groups, H, W, W)
        if hip_reg is not None and hip_reg(feat0):
            # nndepth_amd's regulariser on its HIP path reads the rows of the feature volume and writes the rows of the
            # geometry volume straight into its pyramid buffer: no permuted copies either side
            _, _, total = ops.pyramid_layout(B * num_groups, H, W, num_levels)
            self._geo_buf = torch.empty(total, dtype=torch.float32, device=feat0.device)
            regularizer_3d.forward_rows(feat0, features, out=self._geo_buf[:feat0.numel()].view(B, num_groups, H, W, W))
            if self._pooled:
                ops.pyramid_pool_levels_(self._geo_buf, B * num_groups, H, W, num_levels)
            return
        if not self._pooled:
            ops.pyramid_pool_levels_(self._feat_buf, B * num_groups, H, W, num_levels)
            self._pooled = True
        geo = regularizer_3d(feat0.clone().permute(0, 1, 4, 2, 3), features)          # (B, G, W2, H, W1)
        geo0 = geo.permute(0, 1, 3, 4, 2).contiguous().float()                          # (B, G, H, W1, W2)
        self._geo_buf = ops.pyramid_from_level0(geo0.view(-1, W), B * num_groups, H, W, num_levels)

    @property
    def _feat(self):  # the feature pyramid, all levels in place
        return self.pyramids()[0]

    @property
    def _geo(self):
        return self.pyramids()[1]

    def pyramids(self, pooled: bool = True):
        """(feature pyramid, geometry pyramid) buffers in ops.pyramid_layout; pooled=False: only their level 0 is promised."""
        if pooled and not self._pooled:
            B, H, W = self.shape
            ops.pyramid_pool_levels_(self._feat_buf, B * self.num_groups, H, W, self.num_levels)
            ops.pyramid_pool_levels_(self._geo_buf, B * self.num_groups, H, W, self.num_levels)
            self._pooled = True
        return self._feat_buf, self._geo_buf

    @property
    def geo_level0(self) -> torch.Tensor:
        """`geo_aware_cv[0]` without asking for the pooled levels."""
        B, H, W = self.shape
        n = B * self.num_groups * H * W
        return self._geo_buf[:n * W].view(n, 1, W)

    def _views(self, pyr):
        B, H, W = self.shape
        self.pyramids()
        n = B * self.num_groups * H * W
        offs, widths, _ = ops.pyramid_layout(B * self.num_groups, H, W, self.num_levels)
        return [pyr[o:o + n * w].view(n, 1, w) for o, w in zip(offs, widths)]

    def interleaved(self) -> torch.Tensor:
        """Both pyramids in the group-interleaved layout the fused refinement loop gathers from (built on first use)."""
        if getattr(self, "_il", None) is None:
            B, H, W = self.shape
            if ops.igev_interleave_level0_supported(self.num_groups, W, self.num_levels):
                self._il = ops.igev_interleave_level0(self._feat_buf, self._geo_buf, B, self.num_groups, H, W, self.num_levels)
            else:
                self._il = ops.igev_interleave_pyramids(*self.pyramids(), B, self.num_groups, H, W, self.num_levels)
        return self._il

    @property
    def feat_corr_cv(self):
        return self._views(self._feat_buf)

    @property
    def geo_aware_cv(self):
        return self._views(self._geo_buf)

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        return ops.igev_lookup(*self.pyramids(), coords.float(), self.num_groups, self.num_levels, self.radius)

    forward = __call__


class AGCL:
    """Drop-in for the reference's adaptive group correlation layer
    (nndepth/models/cre_stereo/cost_volume.py:7-154): same constructor `(fmap1, fmap2, att=None)` and call
    `(flow, extra_offset, small_patch=False, iter_mode=False) -> (N,36,H,W)`.  The warp / window correlation /
    offset sampling run in HIP (csrc/agcl.hip); `att` — the LoFTR cross attention of the 1/32 stage, an arbitrary
    callable on (N, H*W, C) tokens — is called as given, exactly where the reference calls it."""

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, att=None):
        self.fmap1 = fmap1.float().contiguous()
        self.fmap2 = fmap2.float().contiguous()
        self.att = att
        self._scratch = None  # warped right features of iter mode, reused over the iterations
        self._attended = None  # att(fmap1, fmap2): constant over the iterations of a stage (inference), computed once
        self._nhwc = None  # channels-last copies of the attended maps for the offset-mode kernel (C = 256)

    def __call__(self, flow: torch.Tensor, extra_offset: torch.Tensor, small_patch: bool = False,
                 iter_mode: bool = False) -> torch.Tensor:
        if iter_mode:
            if self._scratch is None:
                self._scratch = torch.empty_like(self.fmap2)
            return ops.agcl_corr_iter(self.fmap1, self.fmap2, flow.float(), small_patch, self._scratch)
        f1, f2 = self.attended()
        if f1.shape[1] == 256:  # the line-per-tap kernel is built for the 256-channel maps of the reference's configs
            if self._nhwc is None:
                self._nhwc = (ops.nchw_to_nhwc(f1), ops.nchw_to_nhwc(f2))
            return ops.agcl_corr_offset(self._nhwc[0], self._nhwc[1], flow.float(), extra_offset.float(), small_patch,
                                        channels_last=True)
        return ops.agcl_corr_offset(f1, f2, flow.float(), extra_offset.float(), small_patch)

    def attended(self):
        """(fmap1, fmap2) after the optional cross attention (cost_volume.py:92-101).  The reference re-evaluates `att`
        on the same two maps in every iteration; at inference the result is the same each time, so it is cached."""
        if self.att is None:
            return self.fmap1, self.fmap2
        if self._attended is None and hasattr(self.att, "forward_maps"):
            a, b = self.att.forward_maps(self.fmap1, self.fmap2)  # map-level API of nndepth_amd's transformer: no transposes
            self._attended = (a.float().contiguous(), b.float().contiguous())
        if self._attended is None:
            N, C, H, W = self.fmap1.shape
            a = self.fmap1.permute(0, 2, 3, 1).reshape(N, H * W, C)
            b = self.fmap2.permute(0, 2, 3, 1).reshape(N, H * W, C)
            a, b = self.att(a, b)
            self._attended = (a.reshape(N, H, W, C).permute(0, 3, 1, 2).float().contiguous(),
                              b.reshape(N, H, W, C).permute(0, 3, 1, 2).float().contiguous())
        return self._attended
