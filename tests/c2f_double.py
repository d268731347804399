"""Test double for the encoder side of Coarse2FineGroupRepViTRAFTStereo.  The reference builds a RepViT backbone, three MobileOne
1x1 projections and two FeatureFusionBlocks (PyTorch modules, not on the hot path: SURVEY §8); `make_c2f(base_cls)` replaces them by
a tiny deterministic conv pyramid with the same interfaces, so that the SAME subclass can be built on the reference's class (golden
generation, oracle/make_golden_c2f.py) and on nndepth_amd.raft_stereo.Coarse2FineRAFTStereoBase (tests): everything downstream of
the encoder side — group correlation, ConvGRU update block, convex upsample, the three-stage cascade — is then the code under test."""
import torch
import torch.nn as nn


class TinyPyramid(nn.Module):
    """frames -> [16 ch @1/4, 32 @1/8, 64 @1/16, 128 @1/32, 256 @1/64]; the model takes [::2][::-1]"""

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 12, 3, 2, 1), nn.ReLU(), nn.Conv2d(12, 16, 3, 2, 1), nn.ReLU())
        self.down = nn.ModuleList([nn.Sequential(nn.Conv2d(c, 2 * c, 3, 2, 1), nn.ReLU()) for c in (16, 32, 64, 128)])

    def forward(self, x):
        out = [self.stem(x)]
        for d in self.down:
            out.append(d(out[-1]))
        return out


class TinyFusion(nn.Module):
    """[coarse (C0 @ 1/4 of the resolution), fine (C1)] -> `out` channels at the fine resolution"""

    def __init__(self, c0, c1, out):
        super().__init__()
        self.conv = nn.Conv2d(c0 + c1, out, 3, 1, 1)

    def forward(self, feats):
        coarse, fine = feats
        coarse = nn.functional.interpolate(coarse, size=fine.shape[-2:], mode="bilinear", align_corners=True)
        return torch.tanh(self.conv(torch.cat([coarse, fine], dim=1)))


def make_c2f(base_cls, **kwargs):
    class TinyC2F(base_cls):
        def __init__(self, **kw):
            super().__init__(**kw)
            # (the reference's constructor has built its own cnet_proj / fusion_blocks by now: replaced)
            self.cnet_proj = self._tiny_cnet_proj()
            self.fusion_blocks = self._tiny_fusion_blocks()

        def _tiny_cnet_proj(self):
            return nn.ModuleList([nn.Conv2d(c, self.context_dim * 2, 1) for c in (256, 64, 64)])

        def _tiny_fusion_blocks(self):
            return nn.ModuleList([TinyFusion(256, 64, 64), TinyFusion(64, 16, 64)])

        def _init_fnet(self, **kw):
            return TinyPyramid()

        def _init_cnet_proj(self):
            return nn.ModuleList()

        def _init_fusion_blocks(self):
            return nn.ModuleList()

    return TinyC2F(**kwargs)
