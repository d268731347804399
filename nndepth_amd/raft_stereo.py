"""RAFT-Stereo with the MI355X-native hot path.

`BaseRAFTStereo` keeps the reference's constructor kwargs (= `BaseRAFTStereoModelConfig.to_dict()`,
nndepth/models/raft_stereo/configs.py:11-23), `state_dict()` keys and `forward(frame1, frame2)
-> List[{"up_disp": Tensor}]` (nndepth/models/raft_stereo/model.py:17-163), so the reference's
inference / evaluate scripts can use it unchanged.  Inside `forward()`:

    encoder + cnet_proj      HIP, ONE C-ABI call (csrc/encoder.hip: nnd_encoder_forward; eval-mode BatchNorm folded).
                             `hip_encoder=False` is the explicit opt-in to PyTorch-ROCm modules for the encoder; there is
                             no silent fallback: an encoder the HIP path does not build (group norm, dropout) raises
                                                                                                    model.py:107-109
    corr pyramid build       HIP  (csrc/corr1d.hip)                     model.py:124
    for iters: lookup -> update block -> coords += delta -> convex upsample
                             HIP, ONE C-ABI call for the whole loop     model.py:130-137
                             (csrc/update_block.hip: nnd_raft_stereo_refine)

The classes are inference-only (eval-mode BatchNorm is folded into the convolutions and no autograd graph is recorded):
`forward()` raises in training mode.

`patch(model)` installs the same kernels behind the three duck-typed seams of an *unmodified*
reference model instance: `corr_fn`, `update_block`, `convex_upsample`.
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import NndError
from .blocks import BasicUpdateBlock
from .cost_volume import CorrBlock1D, GroupCorrBlock1D
from .encoder import BasicEncoder
from .upsample import convex_upsample


def load_weights(model: nn.Module, weights: str, strict_load: bool = True) -> nn.Module:
    """.pth / .safetensors, like nndepth/utils/common.py:8-16."""
    if weights.endswith(".safetensors"):
        from safetensors.torch import load_file
        state = load_file(weights, device="cpu")
    else:
        state = torch.load(weights, map_location="cpu")
    model.load_state_dict(state, strict=strict_load)
    return model


def require_eval(model: nn.Module) -> None:
    """The HIP hot path is inference-only: BatchNorm is folded with its running statistics and nothing records an autograd
    graph, so a model left in training mode would silently compute something else than the reference's training forward."""
    if model.training:
        raise NndError(f"{type(model).__name__} is inference-only on the HIP path: call model.eval() first "
                       "(training-mode BatchNorm / autograd are not built)")


class AutoCalibrate:
    """fp16x2 activation range (include/nndepth_amd.h "fp16x2 activation range", csrc/calib.hip), shared by the model classes.

    The fp16x2 arithmetic carries an activation x as two fp16 pieces of x * 2^xs with xs per layer.  A freshly packed layer has
    xs = 2 (all 22 bits for 0.06 <= |x| < 16376); `calibrate(frame1, frame2)` runs the forward once with every layer recording the
    largest |activation| M it stages and then sets xs so that M * 2^xs lies in [2^10, 2^11): all 22 bits from M down to M * 2^-13,
    absolute error <= M * 2^-36 below, finite results up to 32 * M, inf / NaN in the output beyond (never a silently wrong value).
    With `auto_calibrate` (default) the first `forward()` after the parameters were (re)packed calibrates on its own input — one
    extra forward, once; call `calibrate()` yourself with representative frames to choose the data, or set
    `auto_calibrate = False` to keep the default scales.  `activation_ranges()` reports the valid |x| per update-block layer."""

    auto_calibrate = True

    def calibrate(self, *args, max_passes: int = 4, **kwargs):
        require_eval(self)
        for _ in range(max_passes):
            with ops.calibration() as c, torch.no_grad():
                self._forward(*args, **kwargs)
            if not (c.status & 1):  # bit 0: a layer saw inf / NaN at the scale it had (lowered by 2^12 since): go again
                return self
        raise NndError(f"{type(self).__name__}.calibrate: activations still overflow fp16 after {max_passes} passes "
                       "(non-finite inputs or weights?)")

    def _forward_calibrated(self, *args, **kwargs):
        """`_forward`, calibrating first if an fp16x2 engine on its path still has the default activation scales."""
        if self.arithmetic != "fp16x2" or not self.auto_calibrate:
            return self._forward(*args, **kwargs)
        try:
            with ops.require_calibrated():
                return self._forward(*args, **kwargs)
        except ops.NeedsCalibration:
            self.calibrate(*args, **kwargs)
            return self._forward(*args, **kwargs)

    def activation_ranges(self):
        eng = self.update_block.engine
        return eng.activation_ranges() if eng.packed is not None else {}


def hip_encoder_blocker(fnet: nn.Module, norms) -> Optional[str]:
    """Why csrc/encoder.hip cannot run this BasicEncoder, or None."""
    if fnet.norm_fn not in norms:
        return f"norm_fn={fnet.norm_fn!r} is not built in HIP (built: {', '.join(norms)})"
    if fnet.dropout is not None:
        return "dropout > 0"
    return None


class BaseRAFTStereo(AutoCalibrate, nn.Module):
    def __init__(self, iters: int = 12, fnet_dim: int = 256, hidden_dim: int = 128, context_dim: int = 128,
                 corr_levels: int = 4, corr_radius: int = 4, tracing: bool = False,
                 include_preprocessing: bool = False, weights: Optional[str] = None, strict_load: bool = True,
                 fused_loop: bool = True, hip_encoder: bool = True, arithmetic: str = "fp16x2", **kwargs):
        super().__init__()
        self.arithmetic = arithmetic  # update-block / encoder convolutions: "fp16x2" (default; 2 fp16 pieces, parity-gated), "bf16x3" (3 bf16 pieces) or "fp32" (exact fp32 MFMA)
        self.iters, self.fnet_dim, self.hidden_dim, self.context_dim = iters, fnet_dim, hidden_dim, context_dim
        self.corr_levels, self.corr_radius = corr_levels, corr_radius
        self.tracing, self.include_preprocessing = tracing, include_preprocessing
        self.fused_loop = fused_loop
        self.hip_encoder = hip_encoder
        self._enc_engine, self._enc_version = None, None
        self.fnet = BasicEncoder(output_dim=fnet_dim)
        self.cnet_proj = nn.Sequential(nn.Conv2d(fnet_dim, context_dim + hidden_dim, 3, padding=1), nn.ReLU(False))
        self.update_block = BasicUpdateBlock(hidden_dim=hidden_dim, cor_planes=corr_levels * (2 * corr_radius + 1),
                                             flow_channel=1, context_dim=context_dim, spatial_scale=8, arithmetic=arithmetic)
        self.corr_fn = CorrBlock1D
        self.weights, self.strict_load = weights, strict_load
        if weights is not None:
            load_weights(self, weights, strict_load)

    # seams kept as methods so callers that monkey-patch them keep working
    def convex_upsample(self, flow, mask, rate=8):
        return convex_upsample(flow, mask, rate)

    def initialize_coords(self, fmap1):
        B, _, H, W = fmap1.shape
        return torch.arange(W, device=fmap1.device).float()[None, None, None, :].repeat(B, 1, H, 1)

    def _encoder_engine(self, device) -> "ops.EncoderEngine":
        """(Re)pack fnet + cnet_proj for the HIP encoder if their parameters / buffers changed."""
        tensors = list(self.fnet.state_dict().values()) + list(self.cnet_proj.state_dict().values())
        v = (tuple((t.data_ptr(), t._version) for t in tensors), str(device))
        if v != self._enc_version:
            if self._enc_engine is None:
                self._enc_engine = ops.EncoderEngine(self.fnet_dim, self.fnet.norm_fn, self.context_dim + self.hidden_dim,
                                                     self.arithmetic)
            self._enc_engine.load(self.fnet.state_dict(), self.cnet_proj.state_dict(), eps=1e-5, device=device)
            self._enc_version = v
        return self._enc_engine

    def forward_fnet(self, frame1, frame2):
        if self.hip_encoder:
            why = hip_encoder_blocker(self.fnet, ("batch", "none"))
            if why:
                raise NndError(f"BaseRAFTStereo: the HIP encoder cannot run this fnet ({why}); pass hip_encoder=False to run "
                               "the encoder's PyTorch-ROCm modules explicitly")
            # BasicEncoder + cnet_proj in ONE C-ABI call (csrc/encoder.hip); eval-mode BatchNorm folded into the convs
            B = frame1.shape[0]
            # (the two frame tensors are read where they lie: no torch.cat copy, basic_encoder.py:74-76)
            fmaps, cnet = self._encoder_engine(frame1.device).forward(frame1.float(), n_cnet=B, frames_b=frame2.float())
            return fmaps[:B], fmaps[B:], cnet
        fmap1, fmap2 = self.fnet([frame1, frame2])  # explicit opt-in (hip_encoder=False): PyTorch-ROCm modules
        return fmap1, fmap2, self.cnet_proj(fmap1)

    def forward(self, frame1: torch.Tensor, frame2: torch.Tensor, **kwargs) -> List[Dict[str, torch.Tensor]]:
        require_eval(self)
        with torch.no_grad():
            return self._forward_calibrated(frame1, frame2)

    def _forward(self, frame1: torch.Tensor, frame2: torch.Tensor) -> List[Dict[str, torch.Tensor]]:
        fmap1, fmap2, cnet = self.forward_fnet(frame1, frame2)
        rate = frame1.shape[-1] // fmap1.shape[-1]
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        net, inp = ops.split_tanh_relu(cnet.float(), self.hidden_dim)  # split + tanh + relu in one kernel (model.py:119-122)
        corr = self.corr_fn(fmap1, fmap2, self.corr_levels, self.corr_radius)
        if self.fused_loop and isinstance(corr, CorrBlock1D):
            eng = self.update_block.sync_engine(frame1.device)
            up, _, _ = eng.refine(corr._pyr, self.corr_levels, self.corr_radius, net.float(), inp.float(),
                                  rate, self.iters, keep_all=True)
            return [{"up_disp": up[i]} for i in range(self.iters)]
        # seam-by-seam loop (same shape as the reference's), every step still a HIP kernel
        coords1 = self.initialize_coords(fmap1)
        org = self.initialize_coords(fmap1)
        outs = []
        for _ in range(self.iters):
            sampled = corr(coords1)
            net, mask, delta = self.update_block(net, inp, sampled, coords1 - org)
            coords1 = coords1 + delta
            outs.append({"up_disp": self.convex_upsample(coords1 - org, mask, rate=rate)})
        return outs


class Coarse2FineRAFTStereoBase(AutoCalibrate, nn.Module):
    """The hot path of `Coarse2FineGroupRepViTRAFTStereo` (nndepth/models/raft_stereo/model.py:166-320) on HIP: per cascade stage the
    group correlation pyramid (GroupCorrBlock1D, quirks Q4 / Q6 kept), the ConvGRU update block (`gru="conv_gru"`, spatial scale
    (4, 4)), the convex upsample and the loop around them, stage after stage (1/64 -> 1/16 -> 1/4 with the reference's encoder), every
    iteration's disparity brought to frame size like the reference (`interpolate` x rate, nearest).

    The encoder side is NOT part of the path (the RepViT backbone, the MobileOne `cnet_proj` and the FeatureFusionBlocks
    are PyTorch modules in the reference and stay PyTorch-ROCm modules here): a subclass supplies them through `_init_fnet`,
    `_init_cnet_proj`, `_init_fusion_blocks` with the reference's interfaces — `fnet(x)` returns the feature pyramid of which
    `[::2][::-1]` are the stages, `cnet_proj[idx](fmap1)` -> (B, 2 * context_dim, H, W), split in halves into net / inp,
    `fusion_blocks[idx-1]([previous_feat, feat])` —, or `patch_coarse2fine(model)` adopts them from a reference instance."""

    def __init__(self, iters: int = 12, hidden_dim: int = 128, context_dim: int = 128, corr_levels: int = 1, corr_radius: int = 4,
                 num_groups: int = 4, weights: Optional[str] = None, strict_load: bool = True, fused_loop: bool = True,
                 arithmetic: str = "fp16x2", **kwargs):
        super().__init__()
        assert corr_levels == 1, "Corr level must be 1 in Coarse2FineGroupRepViTRaftStereo"  # model.py:214
        self.arithmetic = arithmetic
        self.iters, self.hidden_dim, self.context_dim = iters, hidden_dim, context_dim
        self.corr_levels, self.corr_radius, self.num_groups = corr_levels, corr_radius, num_groups
        self.fused_loop = fused_loop
        self.fnet = self._init_fnet()
        self.cnet_proj = self._init_cnet_proj()
        self.fusion_blocks = self._init_fusion_blocks()
        self.update_block = BasicUpdateBlock(hidden_dim=hidden_dim, cor_planes=num_groups * corr_levels * (2 * corr_radius + 1),
                                             flow_channel=1, context_dim=context_dim, gru="conv_gru", spatial_scale=(4, 4),
                                             arithmetic=arithmetic)
        self.corr_fn = GroupCorrBlock1D
        self.weights, self.strict_load = weights, strict_load
        if weights is not None:
            load_weights(self, weights, strict_load)

    def _init_fnet(self) -> nn.Module:
        raise NotImplementedError("Coarse2FineRAFTStereoBase: supply the encoder (reference: nndepth.encoders.rep_vit.RepViT)")

    def _init_cnet_proj(self) -> nn.ModuleList:
        raise NotImplementedError("Coarse2FineRAFTStereoBase: supply cnet_proj (reference: three 1x1 MobileOneBlocks, model.py:216-222)")

    def _init_fusion_blocks(self) -> nn.ModuleList:
        raise NotImplementedError("Coarse2FineRAFTStereoBase: supply fusion_blocks (reference: two FeatureFusionBlocks, model.py:223-228)")

    def convex_upsample(self, flow, mask, rate=(4, 4)):
        rate = rate if isinstance(rate, int) else rate[0]
        return convex_upsample(flow, mask, rate)

    def initialize_coords(self, fmap1):
        B, _, H, W = fmap1.shape
        return torch.arange(W, device=fmap1.device).float()[None, None, None, :].repeat(B, 1, H, 1)

    def forward_features(self, frame1: torch.Tensor, frame2: torch.Tensor):
        """Encoder side (PyTorch-ROCm): per stage the fused feature map of both frames (2B, C, H, W) and cnet (model.py:275-288)."""
        B = frame1.shape[0]
        features = self.fnet(torch.cat([frame1, frame2], dim=0))[::2][::-1]
        feats, cnets, previous = [], [], None
        for idx, feat in enumerate(features):
            if previous is not None:
                feat = self.fusion_blocks[idx - 1]([previous, feat])
            feats.append(feat)
            cnets.append(self.cnet_proj[idx](feat[:B].clone()))
            previous = feat
        return feats, cnets

    def forward(self, frame1: torch.Tensor, frame2: torch.Tensor, **kwargs) -> List[Dict[str, torch.Tensor]]:
        require_eval(self)
        with torch.no_grad():
            return self._forward_calibrated(frame1, frame2)

    def _forward(self, frame1: torch.Tensor, frame2: torch.Tensor) -> List[Dict[str, torch.Tensor]]:
        feats, cnets = self.forward_features(frame1, frame2)
        return self.refine_stages(feats, cnets, tuple(frame1.shape[-2:]))

    def refine_stages(self, feats, cnets, frame_hw) -> List[Dict[str, torch.Tensor]]:
        """The cascade behind the encoder side (model.py:280-320): HIP kernels only."""
        B = feats[0].shape[0] // 2
        outs: List[Dict[str, torch.Tensor]] = []
        up_last = None
        for idx, feat in enumerate(feats):
            fmap1, fmap2 = feat[:B].float().contiguous(), feat[B:].float().contiguous()
            cnet = cnets[idx].float()
            net, inp = ops.split_tanh_relu(cnet, cnet.shape[1] // 2)
            corr = self.corr_fn(fmap1, fmap2, self.corr_levels, self.corr_radius, self.num_groups)
            if up_last is not None and tuple(up_last.shape[-2:]) != tuple(fmap1.shape[-2:]):
                raise NndError(f"Coarse2FineRAFTStereoBase: stage {idx} is {tuple(fmap1.shape[-2:])} but the previous stage's "
                               f"upsampled disparity is {tuple(up_last.shape[-2:])} (frame size not divisible by the pyramid's strides)")
            if self.fused_loop and isinstance(corr, GroupCorrBlock1D):
                eng = self.update_block.sync_engine(fmap1.device)
                up, _, _ = eng.refine_group(corr._pyr, self.num_groups, self.corr_levels, self.corr_radius, net, inp, 4, self.iters,
                                            disp_init=up_last, keep_all=True)
                ups = [up[i] for i in range(self.iters)]
            else:  # seam-by-seam loop (same shape as the reference's), every step still a HIP kernel
                org = self.initialize_coords(fmap1)
                coords1 = org if up_last is None else org + up_last
                ups = []
                for _ in range(self.iters):
                    sampled = corr(coords1)
                    net, mask, delta = self.update_block(net, inp, sampled, coords1 - org)
                    coords1 = coords1 + delta
                    ups.append(self.convex_upsample(coords1 - org, mask, rate=(4, 4)))
            for u in ups:
                rate = frame_hw[1] / u.shape[-1]
                outs.append({"up_disp": u if rate == 1 else nn.functional.interpolate(u, size=tuple(frame_hw)) * rate})
            up_last = ups[-1]
        return outs


def patch_coarse2fine(model: nn.Module, arithmetic: str = "fp16x2", fused_loop: bool = True) -> nn.Module:
    """Swap the HIP hot path into a reference `Coarse2FineGroupRepViTRAFTStereo` instance in place: `patch()` for `update_block` and
    `convex_upsample`, `corr_fn` = GroupCorrBlock1D, and `forward` = the reference's encoder side (its own fnet / fusion_blocks /
    cnet_proj modules) followed by Coarse2FineRAFTStereoBase.refine_stages."""
    patch(model, arithmetic)
    model.corr_fn = GroupCorrBlock1D
    model.convex_upsample = lambda flow, mask, rate=(4, 4): convex_upsample(flow, mask, rate if isinstance(rate, int) else rate[0])
    model.arithmetic, model.fused_loop = arithmetic, fused_loop
    cls = Coarse2FineRAFTStereoBase

    def forward(frame1, frame2, **kwargs):
        require_eval(model)
        with torch.no_grad():
            feats, cnets = cls.forward_features(model, frame1, frame2)
            return cls.refine_stages(model, feats, cnets, tuple(frame1.shape[-2:]))

    model.forward = forward
    return model


STEREO_MODELS = {"base-raft-stereo": BaseRAFTStereo}


def patch(model: nn.Module, arithmetic: str = "fp16x2") -> nn.Module:
    """Swap the HIP hot path into a reference-style RAFT-Stereo instance in place:
    `corr_fn`, `update_block` (state_dict carried over) and `convex_upsample`.  `arithmetic`: as on the model classes."""
    old = model.update_block
    sd = old.state_dict()
    hid = sd["flow_head.conv1.weight"].shape[0]
    gin = sd["gru.convz1.weight"].shape[1]
    new = BasicUpdateBlock(hidden_dim=hid, cor_planes=sd["encoder.convc1.weight"].shape[1],
                           context_dim=gin - 2 * hid,
                           gru="sep_conv" if "gru.convz2.weight" in sd else "conv_gru",
                           flow_channel=sd["flow_head.conv2.weight"].shape[0],
                           spatial_scale=int(round((sd["mask.2.weight"].shape[0] // 9) ** 0.5)), arithmetic=arithmetic)
    new.load_state_dict(sd, strict=True)
    new.to(next(old.parameters()).device)
    model.update_block = new
    model.corr_fn = CorrBlock1D
    model.convex_upsample = lambda flow, mask, rate=8: convex_upsample(flow, mask, rate)
    return model


def calibrate_patched(model: nn.Module, frame1: torch.Tensor, frame2: torch.Tensor, max_passes: int = 4) -> nn.Module:
    """fp16x2 activation-range calibration of a `patch()`ed reference model on one pair (see AutoCalibrate): the reference's own
    forward runs inside `ops.calibration()`, so the seams record what they stage."""
    for _ in range(max_passes):
        with ops.calibration() as c, torch.no_grad():
            model(frame1, frame2)
        if not (c.status & 1):
            return model
    raise NndError("calibrate_patched: activations still overflow fp16")
