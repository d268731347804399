"""ORACLE — test infrastructure only.  Not shipped, never on the product path.

Plain-PyTorch fp32 CPU restatement of the reference's RAFT-Stereo inference forward and
of each hot-path piece, written functionally over a `state_dict` (no nn.Module mirror).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Pinned (see tests/test_oracle_golden.py): every function below is checked against golden
vectors produced by the *imported* reference (`oracle/make_golden.py`, run in the build
container where /root/reference exists).

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _conv(sd: SD, name: str, x: torch.Tensor, stride=1, padding=0) -> torch.Tensor:
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _bn_eval(sd: SD, name: str, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    # nn.BatchNorm2d in eval mode (scripts call model.eval(): raft_stereo/scripts/inference.py:77)
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, eps)


# ----------------------------------------------------------------------------- encoder
def residual_block(sd: SD, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    """nndepth/blocks/residual_block.py:53-60 — note the 1x1 `downsample`+norm3 shortcut is
    ALWAYS applied.  downsample.1 is the same module object as norm3."""
    y = torch.relu(_bn_eval(sd, p + ".norm1", _conv(sd, p + ".conv1", x, stride=stride, padding=1)))
    y = torch.relu(_bn_eval(sd, p + ".norm2", _conv(sd, p + ".conv2", y, padding=1)))
    s = _bn_eval(sd, p + ".norm3", _conv(sd, p + ".downsample.0", x, stride=stride))
    return torch.relu(s + y)


def basic_encoder(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """nndepth/encoders/basic_encoder.py:71-93 (norm_fn='batch', dropout 0)."""
    x = torch.relu(_bn_eval(sd, p + ".norm1", _conv(sd, p + ".conv1", x, stride=2, padding=3)))
    for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = residual_block(sd, f"{p}.{layer}.0", x, stride)
        x = residual_block(sd, f"{p}.{layer}.1", x, 1)
    return _conv(sd, p + ".conv2", x)


# ------------------------------------------------------------------- correlation volume
def corr1d_build(fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int) -> List[torch.Tensor]:
    """nndepth/models/raft_stereo/cost_volume.py:12-34,55-61.
    Returns num_levels+1 tensors of shape (B*H*W1, 1, W2_l) (the last one is never read: Q1)."""
    C = fmap1.shape[1]
    a = fmap1.permute(0, 2, 3, 1)
    b = fmap2.permute(0, 2, 1, 3)
    corr = torch.matmul(a, b) / C ** 0.5
    B, H, W1, W2 = corr.shape
    corr = corr.reshape(B * H * W1, 1, W2)
    pyr = [corr]
    for _ in range(num_levels):
        corr = F.avg_pool1d(corr, 2)
        pyr.append(corr)
    return pyr


def linear_sampler(row: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """nndepth/models/raft_stereo/utils.py:4-27 — border clamp (Q2), floor/ceil gather."""
    w2 = row.shape[1]
    x = torch.clamp(x / (w2 - 1), 0, 1) * (w2 - 1)
    i0 = x.floor().long()
    i1 = x.ceil().long()
    v0 = row.gather(1, i0)
    v1 = row.gather(1, i1)
    coef = i1 - x
    return coef * v0 + (1 - coef) * v1


def corr1d_lookup(pyr: Sequence[torch.Tensor], coords: torch.Tensor, num_levels: int, radius: int) -> torch.Tensor:
    """nndepth/models/raft_stereo/cost_volume.py:36-53. coords (B,1,H,W) -> (B, L*(2r+1), H, W)."""
    B, _, H, W = coords.shape
    outs = []
    dx = torch.linspace(-radius, radius, 2 * radius + 1).view(1, -1)
    for i in range(num_levels):
        row = pyr[i].reshape(B * H * W, -1)
        x = dx + coords.reshape(B * H * W, 1) / 2 ** i
        outs.append(linear_sampler(row, x).view(B, H, W, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()


# ----------------------------------------------------------- IGEV geometry-encoding volume
def group_corr_volume(fmap1: torch.Tensor, fmap2: torch.Tensor, num_groups: int) -> torch.Tensor:
    """nndepth/models/igev_stereo/cost_volume.py:81-98 — torch.split(fmap, num_groups) gives chunks of `num_groups`
    CHANNELS; only the first `num_groups` chunks are correlated (Q4).  -> (B, G, H, W1, W2)."""
    g1 = torch.split(fmap1, num_groups, dim=1)
    g2 = torch.split(fmap2, num_groups, dim=1)
    vols = []
    for i in range(num_groups):
        a, b = g1[i].permute(0, 2, 3, 1), g2[i].permute(0, 2, 1, 3)
        vols.append(torch.matmul(a, b) / a.shape[-1] ** 0.5)
    return torch.stack(vols, dim=1)


def igev_pyramids(feat_vol: torch.Tensor, geo_vol: torch.Tensor, num_levels: int):
    """igev_stereo/cost_volume.py:40-52.  feat_vol (B,G,H,W1,W2); geo_vol (B,G,W2,H,W1) as the regulariser returns
    it.  -> two lists of num_levels+1 tensors (B*G*H*W1, 1, W2_l)."""
    B, G, H, W1, W2 = feat_vol.shape
    f = feat_vol.reshape(B * G * H * W1, 1, W2)
    g = geo_vol.permute(0, 1, 3, 4, 2).reshape(B * G * H * W1, 1, W2)
    fp, gp = [f], [g]
    for _ in range(num_levels):
        f, g = F.avg_pool1d(f, 2), F.avg_pool1d(g, 2)
        fp.append(f)
        gp.append(g)
    return fp, gp


def igev_lookup(fp, gp, coords: torch.Tensor, num_groups: int, num_levels: int, radius: int) -> torch.Tensor:
    """igev_stereo/cost_volume.py:54-79 -> (B, L*2*G*(2r+1), H, W), channel = i*2GT + v*GT + g*T + k."""
    B, _, H, W = coords.shape
    outs = []
    dx = torch.linspace(-radius, radius, 2 * radius + 1).reshape(1, -1)
    for i in range(num_levels):
        c = coords.permute(0, 2, 3, 1).unsqueeze(1).repeat(1, num_groups, 1, 1, 1)
        x = c.reshape(B * num_groups * H * W, 1) / 2 ** i + dx
        for pyr in (fp, gp):
            s = linear_sampler(pyr[i].reshape(B * num_groups * H * W, -1), x)
            outs.append(s.reshape(B, num_groups, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()


# ------------------------------------------- GroupCorrBlock1D (Coarse2FineGroupRepViTRAFTStereo)
def raft_group_corr_build(fmap1: torch.Tensor, fmap2: torch.Tensor, num_groups: int, num_levels: int) -> List[torch.Tensor]:
    """nndepth/models/raft_stereo/cost_volume.py:84-92,115-128: chunks of `num_groups` channels, the first `num_groups` chunks,
    each divided by sqrt(C_TOTAL) (Q4) -> num_levels+1 tensors (B*G*H*W1, 1, W2_l), rows ordered (b,g,h,w1)."""
    C = fmap1.shape[1]
    g1 = torch.split(fmap1, num_groups, dim=1)
    g2 = torch.split(fmap2, num_groups, dim=1)
    vols = []
    for i in range(num_groups):
        a, b = g1[i].permute(0, 2, 3, 1), g2[i].permute(0, 2, 1, 3)
        vols.append(torch.matmul(a, b) / C ** 0.5)
    corr = torch.stack(vols, dim=1)
    B, G, H, W1, W2 = corr.shape
    corr = corr.reshape(B * G * H * W1, 1, W2)
    pyr = [corr]
    for _ in range(num_levels):
        corr = F.avg_pool1d(corr, 2)
        pyr.append(corr)
    return pyr


def raft_group_corr_lookup(pyr: Sequence[torch.Tensor], coords: torch.Tensor, num_groups: int, num_levels: int, radius: int) -> torch.Tensor:
    """nndepth/models/raft_stereo/cost_volume.py:94-113.  The (B*G*H*W, 2r+1) samples are viewed as (B, H, W, -1) WITHOUT moving the
    group axis (Q6): kept as is.  coords (B,1,H,W) -> (B, L*G*(2r+1), H, W)."""
    B, _, H, W = coords.shape
    outs = []
    dx = torch.linspace(-radius, radius, 2 * radius + 1).view(1, -1)
    for i in range(num_levels):
        row = pyr[i].reshape(B * num_groups * H * W, -1)
        c = coords.permute(0, 2, 3, 1).unsqueeze(1).repeat(1, num_groups, 1, 1, 1)
        x = dx + c.reshape(B * num_groups * H * W, 1) / 2 ** i
        outs.append(linear_sampler(row, x).view(B, H, W, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def coarse2fine_refine(sd: SD, feats: Sequence[torch.Tensor], cnets: Sequence[torch.Tensor], frame_hw, iters: int, num_groups: int = 4,
                       corr_levels: int = 1, corr_radius: int = 4):
    """The cascade of Coarse2FineGroupRepViTRAFTStereo.forward (nndepth/models/raft_stereo/model.py:272-320) behind its encoder side:
    feats[idx] = the (fused) feature map of stage idx for both frames, (2B, C, H_idx, W_idx); cnets[idx] = cnet_proj[idx](fmap1).
    Update block: conv_gru, spatial_scale (4, 4).  -> list of up_disp at frame size (len = stages * iters)."""
    B = feats[0].shape[0] // 2
    H0, W0 = feats[0].shape[-2:]
    org = torch.arange(W0).float()[None, None, None, :].repeat(B, 1, H0, 1)
    init = org.clone()
    outs = []
    up = None
    for idx, feat in enumerate(feats):
        fmap1, fmap2 = torch.split(feat, [B, B], dim=0)
        cnet = cnets[idx]
        net, inp = torch.split(cnet, cnet.shape[1] // 2, dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        pyr = raft_group_corr_build(fmap1.float(), fmap2.float(), num_groups, corr_levels)
        coords1 = init
        for _ in range(iters):
            samp = raft_group_corr_lookup(pyr, coords1, num_groups, corr_levels, corr_radius)
            net, mask, delta = update_block(sd, "update_block", net, inp, samp, coords1 - org, gru="conv_gru")
            coords1 = coords1 + delta
            up = convex_upsample(coords1 - org, mask, 4)
            rate = frame_hw[1] / up.shape[-1]
            outs.append(up if rate == 1 else F.interpolate(up, size=tuple(frame_hw)) * rate)
        if idx < len(feats) - 1:
            Hn, Wn = feats[idx + 1].shape[-2:]
            org = torch.arange(Wn).float()[None, None, None, :].repeat(B, 1, Hn, 1)
            init = org + up
    return outs


# ------------------------------------------------------------------------- update block
def motion_encoder(sd: SD, p: str, flow: torch.Tensor, corr: torch.Tensor) -> torch.Tensor:
    """nndepth/blocks/update_block.py:57-65."""
    cor = torch.relu(_conv(sd, p + ".convc1", corr))
    cor = torch.relu(_conv(sd, p + ".convc2", cor, padding=1))
    flo = torch.relu(_conv(sd, p + ".convf1", flow, padding=3))
    flo = torch.relu(_conv(sd, p + ".convf2", flo, padding=1))
    out = torch.relu(_conv(sd, p + ".conv", torch.cat([cor, flo], 1), padding=1))
    return torch.cat([out, flow], 1)


def sep_conv_gru(sd: SD, p: str, h: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """nndepth/blocks/gru.py:22-37: (1x5) pass then (5x1) pass."""
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(sd, p + ".convz" + sfx, hx, padding=pad))
        r = torch.sigmoid(_conv(sd, p + ".convr" + sfx, hx, padding=pad))
        q = torch.tanh(_conv(sd, p + ".convq" + sfx, torch.cat([r * h, x], 1), padding=pad))
        h = (1 - z) * h + z * q
    return h


def conv_gru(sd: SD, p: str, h: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """nndepth/blocks/gru.py:53-61 (single 3x3)."""
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(_conv(sd, p + ".convz1", hx, padding=1))
    r = torch.sigmoid(_conv(sd, p + ".convr1", hx, padding=1))
    q = torch.tanh(_conv(sd, p + ".convq1", torch.cat([r * h, x], 1), padding=1))
    return (1 - z) * h + z * q


def update_block(sd: SD, p: str, net, inp, corr, flow, gru: str = "sep_conv"):
    """nndepth/blocks/update_block.py:103-112 -> (net, mask, delta_flow)."""
    mf = motion_encoder(sd, p + ".encoder", flow, corr)
    x = torch.cat([inp, mf], 1)
    net = (sep_conv_gru if gru == "sep_conv" else conv_gru)(sd, p + ".gru", net, x)
    d = _conv(sd, p + ".flow_head.conv2", torch.relu(_conv(sd, p + ".flow_head.conv1", net, padding=1)), padding=1)
    m = _conv(sd, p + ".mask.2", torch.relu(_conv(sd, p + ".mask.0", net, padding=1)))
    return net, 0.25 * m, d


# ----------------------------------------------------------------------- convex upsample
def convex_upsample(flow: torch.Tensor, mask: torch.Tensor, rate: int) -> torch.Tensor:
    """nndepth/models/raft_stereo/model.py:93-105 generalised to C flow channels
    (cre_stereo/model.py:110-122 is the same with C=2)."""
    N, C, H, W = flow.shape
    m = torch.softmax(mask.view(N, 1, 9, rate, rate, H, W), dim=2)
    u = F.unfold(rate * flow, (3, 3), padding=1).view(N, C, 9, 1, 1, H, W)
    u = torch.sum(m * u, dim=2).permute(0, 1, 4, 2, 5, 3)
    return u.reshape(N, C, rate * H, rate * W)


# ------------------------------------------------------------------------ full forward
def raft_stereo_forward(sd: SD, frame1: torch.Tensor, frame2: torch.Tensor, iters: int,
                        hidden_dim: int = 128, context_dim: int = 64,
                        corr_levels: int = 4, corr_radius: int = 4,
                        return_lowres: bool = False):
    """nndepth/models/raft_stereo/model.py:111-139,160-163 (BaseRAFTStereo).
    Returns list of up_disp (one per iteration) [and the 1/8-res disparities]."""
    f = basic_encoder(sd, "fnet", torch.cat([frame1, frame2], 0))
    fmap1, fmap2 = torch.split(f, f.shape[0] // 2, dim=0)
    cnet = torch.relu(_conv(sd, "cnet_proj.0", fmap1, padding=1))
    rate = frame1.shape[-1] // fmap1.shape[-1]
    fmap1, fmap2 = fmap1.float(), fmap2.float()
    net, inp = torch.split(cnet, [hidden_dim, context_dim], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)
    pyr = corr1d_build(fmap1, fmap2, corr_levels)
    B, _, H, W = fmap1.shape
    org = torch.arange(W).float()[None, None, None, :].repeat(B, 1, H, 1)
    coords1 = org.clone()
    ups, lows = [], []
    for _ in range(iters):
        samp = corr1d_lookup(pyr, coords1, corr_levels, corr_radius)
        net, mask, delta = update_block(sd, "update_block", net, inp, samp, coords1 - org)
        coords1 = coords1 + delta
        disp = coords1 - org
        ups.append(convex_upsample(disp, mask, rate))
        lows.append(disp)
    return (ups, lows) if return_lowres else ups


def igev_refine(sd: SD, p: str, fp, gp, net, inp, init_disp, iters: int, num_groups=8, num_levels=4, radius=4, rate=4,
                return_lowres: bool = False):
    """nndepth/models/igev_stereo/model.py:148-158: coords1 = arange + init; the update block and the upsample take the
    ABSOLUTE coords1 (Q5).  -> list of up (B,1,rate*H,rate*W) [, list of the low-resolution coords1 after each iteration]."""
    B, _, H, W = net.shape
    coords1 = torch.arange(W).float()[None, None, None, :].repeat(B, 1, H, 1) + init_disp
    ups, lows = [], []
    for _ in range(iters):
        samp = igev_lookup(fp, gp, coords1, num_groups, num_levels, radius)
        net, mask, delta = update_block(sd, p, net, inp, samp, coords1)
        coords1 = coords1 + delta
        lows.append(coords1)
        ups.append(convex_upsample(coords1, mask, rate))
    return (ups, lows) if return_lowres else ups


def _cbr3d(sd: SD, p: str, x: torch.Tensor, stride: int = 1, upsample: bool = False) -> torch.Tensor:
    """ConvBn3D / Upsampler3D: nndepth/models/igev_stereo/cost_volume.py:101-130 — [trilinear x2, align_corners=True ->]
    Conv3d(3, bias=False) -> BatchNorm3d (eval) -> LeakyReLU(0.01)."""
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="trilinear", align_corners=True)
    y = F.conv3d(x, sd[p + ".conv.weight"], None, stride=stride, padding=1)
    y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"], sd[p + ".bn.bias"],
                     False, 0.0, 1e-5)
    return F.leaky_relu(y, 0.01)


def _feat_gate(sd: SD, p: str, cv: torch.Tensor, feat: torch.Tensor) -> torch.Tensor:
    """FeatureGuidedBlock: igev_stereo/cost_volume.py:133-147 (1x1 conv, BatchNorm2d eval, ReLU, 1x1 conv, sigmoid gate
    broadcast over the candidate axis)."""
    a = torch.relu(_bn_eval(sd, p + ".feat_att.1", _conv(sd, p + ".feat_att.0", feat)))
    return torch.sigmoid(_conv(sd, p + ".feat_att.3", a).unsqueeze(2)) * cv


def cost_volume_filter(sd: SD, p: str, x: torch.Tensor, feats: Sequence[torch.Tensor]) -> torch.Tensor:
    """CostVolumeFilterNetwork.forward: igev_stereo/cost_volume.py:192-210.  x (B,G,W2,H,W1), feats = guide maps at 1/2,
    1/4, 1/8 of (H,W1) with 40/80/160 channels -> regularised volume, same shape as x."""
    c1 = _cbr3d(sd, p + ".conv1.1", _cbr3d(sd, p + ".conv1.0", x, 2))
    c1 = _feat_gate(sd, p + ".conv1_feat_guided", c1, feats[0])
    c2 = _cbr3d(sd, p + ".conv2.1", _cbr3d(sd, p + ".conv2.0", c1, 2))
    c2 = _feat_gate(sd, p + ".conv2_feat_guided", c2, feats[1])
    c3 = _cbr3d(sd, p + ".conv3.1", _cbr3d(sd, p + ".conv3.0", c2, 2))
    c3 = _feat_gate(sd, p + ".conv3_feat_guided", c3, feats[2])
    c2 = _cbr3d(sd, p + ".proj_3", torch.cat((_cbr3d(sd, p + ".conv3_up", c3, upsample=True), c2), 1))
    c2 = _feat_gate(sd, p + ".conv3_up_feat_guided", c2, feats[1])
    c1 = _cbr3d(sd, p + ".proj_2", torch.cat((_cbr3d(sd, p + ".conv2_up", c2, upsample=True), c1), 1))
    c1 = _feat_gate(sd, p + ".conv2_up_feat_guided", c1, feats[0])
    return _cbr3d(sd, p + ".final_conv", _cbr3d(sd, p + ".conv1_up", c1, upsample=True))


def cost_volume_filter_spec(p: str, c: int = 8, feat_channels=(40, 80, 160)):
    """state_dict keys / shapes of CostVolumeFilterNetwork(c, feat_channels), in registration order
    (igev_stereo/cost_volume.py:150-190)."""
    spec = []

    def cbr(name, ci, co):
        spec.append((f"{name}.conv.weight", (co, ci, 3, 3, 3)))
        for k in ("weight", "bias", "running_mean", "running_var"):
            spec.append((f"{name}.bn.{k}", (co,)))
        spec.append((f"{name}.bn.num_batches_tracked", ()))

    def gate(name, cvc, fc):
        spec.append((f"{name}.feat_att.0.weight", (fc // 2, fc, 1, 1)))
        spec.append((f"{name}.feat_att.0.bias", (fc // 2,)))
        for k in ("weight", "bias", "running_mean", "running_var"):
            spec.append((f"{name}.feat_att.1.{k}", (fc // 2,)))
        spec.append((f"{name}.feat_att.1.num_batches_tracked", ()))
        spec.append((f"{name}.feat_att.3.weight", (cvc, fc // 2, 1, 1)))
        spec.append((f"{name}.feat_att.3.bias", (cvc,)))

    for i, (ci, co) in enumerate(((c, 2 * c), (2 * c, 4 * c), (4 * c, 8 * c)), start=1):
        cbr(f"{p}.conv{i}.0", ci, co)
        cbr(f"{p}.conv{i}.1", co, co)
        gate(f"{p}.conv{i}_feat_guided", co, feat_channels[i - 1])
    cbr(p + ".conv3_up", 8 * c, 4 * c)
    cbr(p + ".proj_3", 8 * c, 4 * c)
    gate(p + ".conv3_up_feat_guided", 4 * c, feat_channels[1])
    cbr(p + ".conv2_up", 4 * c, 2 * c)
    cbr(p + ".proj_2", 4 * c, 2 * c)
    gate(p + ".conv2_up_feat_guided", 2 * c, feat_channels[0])
    cbr(p + ".conv1_up", 2 * c, c)
    cbr(p + ".final_conv", c, c)
    return spec


def igev_init_disparity(logits: torch.Tensor) -> torch.Tensor:
    """nndepth/models/igev_stereo/model.py:92-95,145-146: logits (B,D,H,W) = squeezed geometry volume ->
    -sum_d d * softmax_d(logits), (B,1,H,W)."""
    dist = F.softmax(logits, dim=1)
    disp = torch.arange(0, logits.shape[1], dtype=dist.dtype).reshape(1, -1, 1, 1)
    return -torch.sum(disp * dist, dim=1, keepdim=True)


# --------------------------------------------------------------------- pre- / post-processing
def preprocess_frame(frame: torch.Tensor, HW) -> torch.Tensor:
    """nndepth/models/raft_stereo/scripts/inference.py:55-60: (3,h,w) float in 0..255 -> (1,3,H,W) in [-1,1]."""
    frame = F.interpolate(frame.unsqueeze(0), tuple(HW), mode="bilinear")
    return (frame - 127.5) / 127.5


def padder_pads(HW, divis_by: int = 8):
    """nndepth/data/dataloaders/utils.py:7-11 -> [left, right, top, bottom]."""
    ht, wd = HW
    pad_ht = (((ht // divis_by) + 1) * divis_by - ht) % divis_by
    pad_wd = (((wd // divis_by) + 1) * divis_by - wd) % divis_by
    return [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]


def padder_pad(x: torch.Tensor, pads) -> torch.Tensor:
    """utils.py:13-15."""
    return F.pad(x, pads, mode="replicate")


def padder_unpad(x: torch.Tensor, pads) -> torch.Tensor:
    """utils.py:17-21."""
    ht, wd = x.shape[-2:]
    return x[..., pads[2]:ht - pads[3], pads[0]:wd - pads[1]]


def eval_criterion(disp_gt: torch.Tensor, disp_pred: torch.Tensor, valid_mask=None, d_threshold=None, max_flow: float = 1000):
    """nndepth/models/raft_stereo/scripts/evaluate.py:48-83."""
    if disp_pred.shape[-2:] != disp_gt.shape[-2:]:
        scale = disp_gt.shape[-1] // disp_pred.shape[-1]
        gt = -F.max_pool2d(-disp_gt, kernel_size=scale) / scale
        gt = F.interpolate(gt, size=disp_pred.shape[-2:])
    else:
        gt = disp_gt
    e = torch.sum((disp_pred - gt) ** 2, dim=1).sqrt()
    valid = torch.sum(gt ** 2, dim=1, keepdim=True).sqrt() < max_flow
    if valid_mask is not None:
        valid = valid & valid_mask
    e = e.view(-1)[valid.view(-1)]
    out = {"epe": e.mean().item()}
    for k, thr in (d_threshold or {}).items():
        out[k] = (e > thr).float().mean().item()
    return out


def epe(disp_gt: torch.Tensor, disp_pred: torch.Tensor, max_flow: float = 1000.0) -> float:
    """nndepth/models/raft_stereo/scripts/evaluate.py:62-83 for equal-size inputs."""
    e = torch.sum((disp_pred - disp_gt) ** 2, dim=1).sqrt()
    mag = torch.sum(disp_gt ** 2, dim=1, keepdim=True).sqrt()
    valid = mag < max_flow
    return e.view(-1)[valid.view(-1)].mean().item()


def raft_stereo_spec(fnet_dim=256, hidden_dim=128, context_dim=64, corr_levels=4, corr_radius=4,
                     flow_channel=1, spatial_scale=8) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) list of BaseRAFTStereo.state_dict() — used to generate weights without
    constructing any module.  Order/keys verified against the imported reference."""
    spec: List[Tuple[str, Tuple[int, ...]]] = []

    def conv(name, co, ci, kh, kw):
        spec.append((name + ".weight", (co, ci, kh, kw)))
        spec.append((name + ".bias", (co,)))

    def bn(name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            spec.append((f"{name}.{s}", (c,)))
        spec.append((name + ".num_batches_tracked", ()))

    bn("fnet.norm1", 64)
    conv("fnet.conv1", 64, 3, 7, 7)
    cin = 64
    for li, dim in (("layer1", 64), ("layer2", 96), ("layer3", 128)):
        for bi in (0, 1):
            p = f"fnet.{li}.{bi}"
            conv(p + ".conv1", dim, cin, 3, 3)
            conv(p + ".conv2", dim, dim, 3, 3)
            bn(p + ".norm1", dim)
            bn(p + ".norm2", dim)
            bn(p + ".norm3", dim)
            conv(p + ".downsample.0", dim, cin, 1, 1)
            bn(p + ".downsample.1", dim)
            cin = dim
    conv("fnet.conv2", fnet_dim, 128, 1, 1)
    conv("cnet_proj.0", hidden_dim + context_dim, fnet_dim, 3, 3)
    spec += update_block_spec("update_block", hidden_dim, corr_levels * (2 * corr_radius + 1),
                              context_dim, flow_channel, spatial_scale)
    return spec


def update_block_spec(p, hidden_dim, cor_planes, context_dim, flow_channel, spatial_scale, gru="sep_conv"):
    spec = []

    def conv(name, co, ci, kh, kw):
        spec.append((name + ".weight", (co, ci, kh, kw)))
        spec.append((name + ".bias", (co,)))

    conv(p + ".encoder.convc1", 256, cor_planes, 1, 1)
    conv(p + ".encoder.convc2", 192, 256, 3, 3)
    conv(p + ".encoder.convf1", 128, flow_channel, 7, 7)
    conv(p + ".encoder.convf2", 64, 128, 3, 3)
    conv(p + ".encoder.conv", hidden_dim - flow_channel, 256, 3, 3)
    cin = hidden_dim + context_dim + hidden_dim
    if gru == "sep_conv":
        for n in ("convz1", "convr1", "convq1"):
            conv(f"{p}.gru.{n}", hidden_dim, cin, 1, 5)
        for n in ("convz2", "convr2", "convq2"):
            conv(f"{p}.gru.{n}", hidden_dim, cin, 5, 1)
    else:
        for n in ("convz1", "convr1", "convq1"):
            conv(f"{p}.gru.{n}", hidden_dim, cin, 3, 3)
    conv(p + ".flow_head.conv1", hidden_dim, hidden_dim, 3, 3)
    conv(p + ".flow_head.conv2", flow_channel, hidden_dim, 3, 3)
    conv(p + ".mask.0", hidden_dim * 2, hidden_dim, 3, 3)
    conv(p + ".mask.2", spatial_scale * spatial_scale * 9, hidden_dim * 2, 1, 1)
    return spec
