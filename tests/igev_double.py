"""Test double for IGEV's backbone.  The reference's `IGEVStereoMBNet` needs timm's pretrained MobileNetV3 (not
available offline), but its `IGEVStereoBase` is an abstract class with three hooks (`_init_fnet`,
`_init_cost_volume_filter`, `forward_fnet`).  `make_igev(base_cls, regulariser_cls)` fills those hooks with a tiny
deterministic conv pyramid so that the SAME subclass can be built on the reference's base class (golden generation,
oracle/make_golden_igev.py) and on nndepth_amd.igev_stereo.IGEVStereoBase (tests): everything downstream of the backbone
— volume, regulariser, soft-argmin init, refinement loop — is then the code under test."""
import torch
import torch.nn as nn


class TinyBackbone(nn.Module):
    """frames -> [24 ch @1/4, 40 ch @1/8, 80 ch @1/16, 160 ch @1/32]"""

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 16, 3, 2, 1), nn.ReLU(), nn.Conv2d(16, 24, 3, 2, 1), nn.ReLU())
        self.d8 = nn.Sequential(nn.Conv2d(24, 40, 3, 2, 1), nn.ReLU())
        self.d16 = nn.Sequential(nn.Conv2d(40, 80, 3, 2, 1), nn.ReLU())
        self.d32 = nn.Sequential(nn.Conv2d(80, 160, 3, 2, 1), nn.ReLU())

    def forward(self, x):
        f4 = self.stem(x)
        f8 = self.d8(f4)
        f16 = self.d16(f8)
        return [f4, f8, f16, self.d32(f16)]


def make_igev(base_cls, regulariser_cls, **kwargs):
    class TinyIGEV(base_cls):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.fnet_proj = nn.Sequential(nn.Conv2d(24, self.hidden_dim * 2, 3, 1, 1), nn.ReLU(False))
            self.cnet_proj = nn.Sequential(nn.Conv2d(24, self.context_dim * 2, 3, 1, 1), nn.ReLU(False))

        def _init_fnet(self):
            return TinyBackbone()

        def _init_cost_volume_filter(self):
            return regulariser_cls(self.cv_groups, [40, 80, 160])

        def forward_fnet(self, frame1, frame2):
            B = frame1.shape[0]
            feats = self.fnet(torch.cat([frame1, frame2], dim=0))
            cnet1 = self.cnet_proj(feats[0][:B].clone())
            fmap1, fmap2 = torch.split(self.fnet_proj(feats[0]), B, dim=0)
            return fmap1, fmap2, cnet1, [f[:B] for f in feats[1:]]

    return TinyIGEV(**kwargs)

