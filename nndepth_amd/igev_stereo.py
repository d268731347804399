"""IGEV-Stereo with the MI355X-native hot path.

`IGEVStereoBase` keeps the reference's constructor kwargs, attribute / `state_dict()` names and
`forward(frame1, frame2) -> List[{"up_disp": (B,1,H,W)}]` of nndepth/models/igev_stereo/model.py:15-158, including its
two hooks for subclasses (`_init_fnet`, `_init_cost_volume_filter`, `forward_fnet`).  Inside `forward()`:

    backbone (`forward_fnet`)                       whatever the subclass provides (PyTorch)
    group-wise correlation volume + pyramids        HIP  csrc/corr1d.hip (GeometryAwareCostVolume)     model.py:133-141
    3-D regulariser (Conv3d hourglass)              HIP  csrc/conv3d.hip: every Conv3d+BN+LeakyReLU one MFMA-conv launch on
                                                    depth-major volumes, trilinear x2 and feature gating kernels
                                                    (`CostVolumeFilterNetwork` below; `.hip = False` keeps PyTorch ops)
    cv_squeezer Conv3d + soft-argmin init           HIP  nnd_igev_init_disparity (one pass over the volume) model.py:144-146
    for iters: combined lookup -> update block -> coords += delta -> convex upsample (absolute coords, Q5)
                                                    HIP, ONE C-ABI call: nnd_igev_stereo_refine          model.py:152-158

`IGEVStereoMBNet` (MobileNetV3 backbone from timm) is declared for interface parity; timm is an external dependency of
the reference that is not part of this repository.
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

import warnings

from . import ops
from ._lib import NndError
from .blocks import BasicUpdateBlock
from .cost_volume import GeometryAwareCostVolume
from .raft_stereo import AutoCalibrate, load_weights, require_eval
from .upsample import convex_upsample


# ------------------------------------------------------------------ a15: 3-D regulariser (HIP on the GPU at inference)
def _cbr3d(cin: int, cout: int, stride: int) -> nn.Module:
    """Conv3d(bias=False) + BatchNorm3d + LeakyReLU(0.01), parameter names conv / bn as in the reference's ConvBn3D
    (nndepth/models/igev_stereo/cost_volume.py:101-115)."""
    m = nn.Module()
    m.conv = nn.Conv3d(cin, cout, 3, stride, 1, bias=False)
    m.bn = nn.BatchNorm3d(cout)
    m.relu = nn.LeakyReLU()
    return m


def _run_cbr(m: nn.Module, x: torch.Tensor, upsample: bool = False) -> torch.Tensor:
    if upsample:  # Upsampler3D (cost_volume.py:118-130): trilinear x2, align_corners=True, then conv-bn-act
        x = F.interpolate(x, scale_factor=2.0, mode="trilinear", align_corners=True)
    return m.relu(m.bn(m.conv(x)))


class _Gate(nn.Module):
    """FeatureGuidedBlock (cost_volume.py:133-147): sigmoid(1x1 conv stack of the guide map) scales every candidate."""

    def __init__(self, cv_channel: int, feat_channel: int):
        super().__init__()
        self.feat_att = nn.Sequential(nn.Conv2d(feat_channel, feat_channel // 2, 1), nn.BatchNorm2d(feat_channel // 2),
                                      nn.ReLU(False), nn.Conv2d(feat_channel // 2, cv_channel, 1))

    def forward(self, cv: torch.Tensor, feat: torch.Tensor) -> torch.Tensor:
        return torch.sigmoid(self.feat_att(feat).unsqueeze(2)) * cv


class CostVolumeFilterNetwork(nn.Module):
    """3-level Conv3d hourglass with feature gating (cost_volume.py:150-210); same parameter names."""

    def __init__(self, in_channels: int, feat_channels: List[int]):
        super().__init__()
        c = in_channels
        for i, (ci, co) in enumerate(((c, 2 * c), (2 * c, 4 * c), (4 * c, 8 * c)), start=1):
            seq = nn.Sequential(_cbr3d(ci, co, 2), _cbr3d(co, co, 1))
            setattr(self, f"conv{i}", seq)
            setattr(self, f"conv{i}_feat_guided", _Gate(co, feat_channels[i - 1]))
        self.conv3_up = _cbr3d(8 * c, 4 * c, 1)
        self.proj_3 = _cbr3d(8 * c, 4 * c, 1)
        self.conv3_up_feat_guided = _Gate(4 * c, feat_channels[1])
        self.conv2_up = _cbr3d(4 * c, 2 * c, 1)
        self.proj_2 = _cbr3d(4 * c, 2 * c, 1)
        self.conv2_up_feat_guided = _Gate(2 * c, feat_channels[0])
        self.conv1_up = _cbr3d(2 * c, c, 1)
        self.final_conv = _cbr3d(c, c, 1)
        self.hip = True  # on the GPU at inference: Conv3d / upsample / gating in HIP (False keeps the PyTorch ops)
        self.arithmetic = "fp32"  # "bf16x3": stride-1 Conv3d layers on the split-bf16 MFMA kernel (csrc/conv_split.hip)
        self._hip, self._hip_version = None, None

    def forward(self, x: torch.Tensor, features: List[torch.Tensor]) -> torch.Tensor:
        if self.hip:  # no silent fallback: `.hip = False` is the explicit opt-in to the PyTorch ops below
            require_eval(self)
            return self._forward_hip(x, features)

        def down(seq, t):
            return _run_cbr(seq[1], _run_cbr(seq[0], t))
        c1 = self.conv1_feat_guided(down(self.conv1, x), features[0])
        c2 = self.conv2_feat_guided(down(self.conv2, c1), features[1])
        c3 = self.conv3_feat_guided(down(self.conv3, c2), features[2])
        c2 = _run_cbr(self.proj_3, torch.cat((_run_cbr(self.conv3_up, c3, True), c2), dim=1))
        c2 = self.conv3_up_feat_guided(c2, features[1])
        c1 = _run_cbr(self.proj_2, torch.cat((_run_cbr(self.conv2_up, c2, True), c1), dim=1))
        c1 = self.conv2_up_feat_guided(c1, features[0])
        return _run_cbr(self.final_conv, _run_cbr(self.conv1_up, c1, True))

    # ---- HIP path (csrc/conv3d.hip): the whole hourglass on depth-major volumes, every Conv3d one MFMA-conv launch per sample
    def _engines(self, device):
        tensors = list(self.state_dict().values())
        v = (tuple((t.data_ptr(), t._version) for t in tensors), str(device), self.arithmetic)
        if v == self._hip_version:
            return self._hip
        def c3(m, stride=1, split=0):
            bn = (m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var)
            return ops.Conv3dNorm(m.conv.weight, None, stride, bn, m.bn.eps, m.relu.negative_slope, split, device, self.arithmetic)
        def gate(g):
            a, bnm, _, b = g.feat_att
            bn = (bnm.weight, bnm.bias, bnm.running_mean, bnm.running_var)
            return (ops.ConvNorm(a.weight, a.bias, 1, bn, bnm.eps, device), ops.ConvNorm(b.weight, b.bias, 1, None, 1e-5, device))
        e = {}
        for i in (1, 2, 3):
            seq = getattr(self, f"conv{i}")
            e[f"conv{i}"] = (c3(seq[0], 2), c3(seq[1]))
            e[f"g{i}"] = gate(getattr(self, f"conv{i}_feat_guided"))
        e["conv3_up"], e["conv2_up"], e["conv1_up"], e["final"] = c3(self.conv3_up), c3(self.conv2_up), c3(self.conv1_up), c3(self.final_conv)
        e["proj_3"] = c3(self.proj_3, split=self.conv3_up.conv.out_channels)
        e["proj_2"] = c3(self.proj_2, split=self.conv2_up.conv.out_channels)
        e["g3u"], e["g2u"] = gate(self.conv3_up_feat_guided), gate(self.conv2_up_feat_guided)
        self._hip, self._hip_version = e, v
        return e

    def forward_rows(self, rows: torch.Tensor, features: List[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same network on a volume kept as the pyramids keep it, (B,G,H,W1,W2) with the candidate axis contiguous
        (`forward(x)` with x = rows.permute(0,1,4,2,3)); the result, in the same layout, optionally written into `out`.
        Used by GeometryAwareCostVolume to skip the permuted copies either side of the regulariser (HIP path only)."""
        return self._forward_hip(rows, features, rows_layout=True, out=out)

    def hip_active(self, x: torch.Tensor) -> bool:
        if self.hip:
            require_eval(self)
        return bool(self.hip)

    def _forward_hip(self, x: torch.Tensor, features: List[torch.Tensor], rows_layout: bool = False, out=None) -> torch.Tensor:
        e = self._engines(x.device)
        feats = [f.float() for f in features]

        def gated(vol, g, feat):
            return ops.volume_gate_(vol, g[1](g[0](feat, relu=True)))

        def down(key, vol):
            a, b = e[key]
            return b(a(vol))
        x0 = ops.volume_rows_to_depth_major(x.float()) if rows_layout else ops.volume_to_depth_major(x.float())
        c1 = gated(down("conv1", x0), e["g1"], feats[0])
        c2 = gated(down("conv2", c1), e["g2"], feats[1])
        c3 = gated(down("conv3", c2), e["g3"], feats[2])
        c2 = gated(e["proj_3"](e["conv3_up"](ops.volume_upsample2x(c3)), c2), e["g3u"], feats[1])
        c1 = gated(e["proj_2"](e["conv2_up"](ops.volume_upsample2x(c2)), c1), e["g2u"], feats[0])
        y = e["final"](e["conv1_up"](ops.volume_upsample2x(c1)))
        return ops.depth_major_to_volume_rows(y, out) if rows_layout else ops.depth_major_to_volume(y)


# ------------------------------------------------------------------ the model
class IGEVStereoBase(AutoCalibrate, nn.Module):
    def __init__(self, update_cls: str = "basic_update_block", cv_groups: int = 8, iters: int = 12, hidden_dim: int = 128,
                 context_dim: int = 128, corr_levels: int = 4, corr_radius: int = 4, tracing: bool = False,
                 include_preprocessing: bool = False, weights: Optional[str] = None, strict_load: bool = True,
                 fused_loop: bool = True, arithmetic: str = "fp16x2"):
        super().__init__()
        self.arithmetic = arithmetic  # update-block / encoder convolutions: "fp16x2" (default; 2 fp16 pieces, parity-gated), "bf16x3" (3 bf16 pieces) or "fp32" (exact fp32 MFMA)
        if update_cls != "basic_update_block":
            raise KeyError(update_cls)
        self.fnet = self._init_fnet()
        self.iters, self.hidden_dim, self.context_dim = iters, hidden_dim, context_dim
        self.cv_groups, self.corr_levels, self.corr_radius = cv_groups, corr_levels, corr_radius
        self.update_block = BasicUpdateBlock(hidden_dim=hidden_dim, context_dim=context_dim, flow_channel=1,
                                             cor_planes=corr_levels * (2 * corr_radius + 1) * cv_groups * 2, spatial_scale=4,
                                             arithmetic=arithmetic)
        self.cv_regularizer = self._init_cost_volume_filter()
        if hasattr(self.cv_regularizer, "arithmetic"):  # the HIP regulariser follows the model's arithmetic (5.4 -> 4.3 ms per
            self.cv_regularizer.arithmetic = arithmetic  # 544x960 sample with fp16x2: profiles/r03_igev_regulariser_layers_*.txt)
        self.corr_fn = GeometryAwareCostVolume
        self.cv_squeezer = nn.Conv3d(cv_groups, 1, 3, 1, 1)
        self.tracing, self.include_preprocessing = tracing, include_preprocessing
        self.weights, self.strict_load = weights, strict_load
        self.fused_loop = fused_loop

    def _squeezer_host(self):
        """Host copy of the cv_squeezer parameters (kernel arguments of nnd_igev_init_disparity), refreshed when they change."""
        w, b = self.cv_squeezer.weight, self.cv_squeezer.bias
        key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))
        if getattr(self, "_sq_cache", (None,))[0] != key:
            self._sq_cache = (key, w.detach().float().cpu().contiguous(), None if b is None else b.detach().float().cpu().contiguous())
        return self._sq_cache[1], self._sq_cache[2]

    def _init_fnet(self):
        raise NotImplementedError("Must be implemented in child class")

    def _init_cost_volume_filter(self):
        raise NotImplementedError("Must be implemented in child class")

    def forward_fnet(self, frame1: torch.Tensor, frame2: torch.Tensor):
        """Must return fmap1, fmap2, cnet1 and guide_features."""
        raise NotImplementedError("Must be implemented in child class")

    def regress_disparity(self, distribution: torch.Tensor, width: int) -> torch.Tensor:
        disp = torch.arange(0, width, dtype=distribution.dtype, device=distribution.device).reshape(1, -1, 1, 1)
        return -torch.sum(disp * distribution, dim=1, keepdim=True)

    def initialize_coords(self, fmap1):
        B, _, H, W = fmap1.shape
        return torch.arange(W, device=fmap1.device).float()[None, None, None, :].repeat(B, 1, H, 1)

    def convex_upsample(self, flow, mask, rate=4):
        return convex_upsample(flow, mask, rate)

    def forward(self, frame1: torch.Tensor, frame2: torch.Tensor, **kwargs) -> List[Dict[str, torch.Tensor]]:
        require_eval(self)
        with torch.no_grad():
            return self._forward_calibrated(frame1, frame2)

    def _forward(self, frame1: torch.Tensor, frame2: torch.Tensor) -> List[Dict[str, torch.Tensor]]:
        fmap1, fmap2, cnet1, guide_features = self.forward_fnet(frame1, frame2)
        fnet_ds = frame1.shape[-1] // fmap1.shape[-1]
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        net, inp = ops.split_tanh_relu(cnet1.float(), cnet1.shape[1] // 2)  # split + tanh + relu in one kernel (model.py:129-131)
        corr = self.corr_fn(fmap1, fmap2, guide_features, self.cv_regularizer, self.corr_levels, self.corr_radius,
                            self.cv_groups)
        B, _, H1, W1 = fmap1.shape
        W2 = fmap2.shape[-1]
        geo0 = corr.geo_level0 if hasattr(corr, "geo_level0") else corr.geo_aware_cv[0]  # model.py:144
        if ops.igev_init_disparity_supported(self.cv_groups, W2):
            # cv_squeezer + softmax + regress_disparity as one kernel over the volume where it lies
            init = ops.igev_init_disparity(geo0, *self._squeezer_host(), B, self.cv_groups, H1, W1, W2)
        else:
            # shapes the fused kernel is not built for (more than 8 groups or 512 candidates): the squeezer runs as a
            # PyTorch-ROCm Conv3d, the soft-argmin in HIP — said once, never silently
            warnings.warn(f"IGEVStereoBase: cv_groups={self.cv_groups} / {W2} candidates are outside the fused squeezer + "
                          "soft-argmin kernel (<= 8 groups, <= 512 candidates); the squeezer Conv3d runs on PyTorch-ROCm",
                          RuntimeWarning, stacklevel=2)
            logits = self.cv_squeezer(geo0.reshape(B, self.cv_groups, H1, W1, W2).permute(0, 1, 4, 2, 3)).squeeze(1)
            init = ops.softargmin_disparity(logits.float())
        if self.fused_loop and isinstance(corr, GeometryAwareCostVolume):
            eng = self.update_block.sync_engine(frame1.device)
            il = corr.interleaved()
            # the loop gathers from the interleaved copy alone when its fused lookup is built for these groups / radius:
            # the pooled levels of the two pyramids are then never made
            feat, geo = corr.pyramids(pooled=not ops.igev_refine_reads_interleaved(self.cv_groups, self.corr_levels, self.corr_radius))
            up, low, _ = eng.refine_igev(feat, geo, self.cv_groups, self.corr_levels, self.corr_radius,
                                         net.float(), inp.float(), fnet_ds, self.iters, disp_init=init, keep_all=True,
                                         interleaved=il)
            self.last_low_coords = low  # the loop's state after the last iteration: absolute coordinates at 1/4 resolution (Q5)
            return [{"up_disp": up[i]} for i in range(self.iters)]
        coords1 = self.initialize_coords(fmap1) + init
        outs = []
        for _ in range(self.iters):
            net, mask, delta = self.update_block(net, inp, corr(coords1), coords1)
            coords1 = coords1 + delta
            outs.append({"up_disp": self.convex_upsample(coords1, mask, rate=fnet_ds)})
        self.last_low_coords = coords1
        return outs


class IGEVStereoMBNet(IGEVStereoBase):
    """IGEV-Stereo with the timm MobileNetV3-Large backbone (nndepth/models/igev_stereo/model.py:163-203)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.fnet_proj = nn.Sequential(nn.Conv2d(24, self.hidden_dim * 2, 3, 1, 1), nn.ReLU(False))
        self.cnet_proj = nn.Sequential(nn.Conv2d(24, self.context_dim * 2, 3, 1, 1), nn.ReLU(False))
        if self.weights is not None:
            load_weights(self, self.weights, self.strict_load)

    def _init_fnet(self):
        try:
            from timm.models.mobilenetv3 import tf_mobilenetv3_large_100
        except ImportError as e:  # the reference pins timm==1.0.16 (docker/requirements.txt); not vendored here
            raise ImportError("IGEVStereoMBNet needs timm's tf_mobilenetv3_large_100 backbone") from e

        class _MobilenetV3LargeEncoder(nn.Module):  # nndepth/encoders/mobilenetv3_encoder.py: stages 1-5 are hooked
            def __init__(self):
                super().__init__()
                self.backbone = tf_mobilenetv3_large_100(pretrained=True, features_only=True)

            def forward(self, x):
                bb = self.backbone
                x = bb.act1(bb.bn1(bb.conv_stem(x)))
                feats = []
                for i, blk in enumerate(bb.blocks):
                    x = blk(x)
                    if i in (1, 2, 3, 4, 5):
                        feats.append(x)
                return feats

        return _MobilenetV3LargeEncoder()

    def _init_cost_volume_filter(self):
        return CostVolumeFilterNetwork(self.cv_groups, [40, 80, 160])

    def forward_fnet(self, frame1: torch.Tensor, frame2: torch.Tensor):
        B = frame1.shape[0]
        feats = self.fnet(torch.cat([frame1, frame2], dim=0))
        fmaps = feats[0]
        cnet1 = self.cnet_proj(fmaps[:B].clone())
        fmap1, fmap2 = torch.split(self.fnet_proj(fmaps), B, dim=0)
        return fmap1, fmap2, cnet1, [feats[i][:B] for i in (1, 2, 4)]
