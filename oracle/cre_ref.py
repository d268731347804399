"""ORACLE — test infrastructure only.  Not shipped, never on the product path.

Plain-PyTorch fp32 CPU restatement of the reference's CREStereo hot path:
the adaptive group correlation layer (both modes, both window shapes), its zero-padded bilinear
sampler, and the 3-scale cascade around them (LoFTR linear attention + sine position encoding +
instance-norm encoder restated as well so that the cascade can be pinned end to end).
Written functionally over a `state_dict`.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it.

Pinned by tests/test_oracle_golden.py against golden vectors produced by the *imported* reference
(`oracle/make_golden_cre.py`, run in the build container where /root/reference exists).
Each function cites the reference lines it restates (paths relative to /root/reference).
"""
import math
from typing import Callable, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .torch_ref import SD, _conv, convex_upsample, update_block, update_block_spec


# ------------------------------------------------------------------------ a19: sampler
def bilinear_sampler(img: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """nndepth/models/cre_stereo/utils.py:5-20,34-107.  img (N,C,H,W); coords (N,Hg,Wg,2) = (x, y) in pixels.
    Pixel coordinates go to [-1,1] and back (align_corners=True) — kept, because the round trip is not exact in
    fp32; then 4 taps on the image as if it were surrounded by zeros."""
    N, C, H, W = img.shape
    x = coords[..., 0]
    y = coords[..., 1]
    x = ((2 * x / (W - 1) - 1) + 1) / 2 * (W - 1)
    y = ((2 * y / (H - 1) - 1) + 1) / 2 * (H - 1)
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    x1 = x0 + 1
    y1 = y0 + 1
    w00 = (x1 - x) * (y1 - y)   # tap (x0, y0)
    w01 = (x1 - x) * (y - y0)   # tap (x0, y1)
    w10 = (x - x0) * (y1 - y)   # tap (x1, y0)
    w11 = (x - x0) * (y - y0)   # tap (x1, y1)
    flat = img.reshape(N, C, H * W)

    def tap(xi, yi):
        ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long().reshape(N, 1, -1).expand(-1, C, -1)
        v = torch.gather(flat, 2, idx)
        return v * ok.reshape(N, 1, -1).to(img.dtype)

    out = (tap(x0, y0) * w00.reshape(N, 1, -1) + tap(x0, y1) * w01.reshape(N, 1, -1)
           + tap(x1, y0) * w10.reshape(N, 1, -1) + tap(x1, y1) * w11.reshape(N, 1, -1))
    return out.reshape(N, C, coords.shape[1], coords.shape[2])


def coords_grid(N: int, H: int, W: int) -> torch.Tensor:
    """utils.py:23-26: channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(N, 1, 1, 1)


def _window(small_patch: bool) -> List[Tuple[int, int]]:
    """(dy, dx) of the 9 search positions: 3x3 (dy outer, dx inner) or 1x9 (cost_volume.py:63-68,41-46)."""
    if small_patch:
        return [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    return [(0, dx) for dx in range(-4, 5)]


# ------------------------------------------------------------------------ a17: corr_iter
def agcl_corr_iter(fmap1: torch.Tensor, fmap2: torch.Tensor, flow: torch.Tensor, small_patch: bool) -> torch.Tensor:
    """cost_volume.py:28-79: warp the right features by coords+flow, then per channel group (4 groups) the mean
    over the group's channels of left * warped-right at the 9 window positions of the REPLICATE-padded warped map."""
    N, C, H, W = fmap1.shape
    coords = (coords_grid(N, H, W) + flow).permute(0, 2, 3, 1)
    warped = bilinear_sampler(fmap2, coords)
    out = []
    G = C // 4
    for g in range(4):
        l = fmap1[:, g * G:(g + 1) * G]
        r = warped[:, g * G:(g + 1) * G]
        for dy, dx in _window(small_patch):
            ys = (torch.arange(H) + dy).clamp(0, H - 1)
            xs = (torch.arange(W) + dx).clamp(0, W - 1)
            out.append(torch.mean(l * r[:, :, ys][:, :, :, xs], dim=1, keepdim=True))
    return torch.cat(out, 1)


# ------------------------------------------------------------------------ a18: corr_att_offset
def agcl_corr_att_offset(fmap1: torch.Tensor, fmap2: torch.Tensor, flow: torch.Tensor, extra_offset: torch.Tensor,
                         small_patch: bool, att: Optional[Callable] = None) -> torch.Tensor:
    """cost_volume.py:81-154: optional cross attention on (N, HW, C) tokens; per group the right features are
    sampled at coords + flow + window offset + learned offset (extra_offset (N,18,H,W) viewed as (N,9,2,H,W):
    channel 2k = x, 2k+1 = y) and correlated (channel mean) with the left features."""
    N, C, H, W = fmap1.shape
    if att is not None:
        a = fmap1.permute(0, 2, 3, 1).reshape(N, H * W, C)
        b = fmap2.permute(0, 2, 3, 1).reshape(N, H * W, C)
        a, b = att(a, b)
        fmap1 = a.reshape(N, H, W, C).permute(0, 3, 1, 2)
        fmap2 = b.reshape(N, H, W, C).permute(0, 3, 1, 2)
    G = C // 4
    eo = extra_offset.reshape(N, 9, 2, H, W)
    base = coords_grid(N, H, W) + flow  # (N,2,H,W)
    win = _window(small_patch)
    out = []
    for g in range(4):
        l = fmap1[:, g * G:(g + 1) * G]
        r = fmap2[:, g * G:(g + 1) * G]
        for k, (dy, dx) in enumerate(win):
            # reference order: offsets = window + extra, then coords + offsets
            sx = base[:, 0] + (float(dx) + eo[:, k, 0])
            sy = base[:, 1] + (float(dy) + eo[:, k, 1])
            s = bilinear_sampler(r, torch.stack([sx, sy], -1))
            out.append(torch.mean(l * s, dim=1, keepdim=True))
    return torch.cat(out, 1)


# ------------------------------------------------------------------------ LoFTR pieces (§8f-4, restated so the cascade is pinned)
def pos_enc_sine(d_model: int, H: int, W: int) -> torch.Tensor:
    """nndepth/blocks/pos_enc.py:22-42 with temp_bug_fix=False: note `-log(1e4) / d_model // 2` (floor division
    applied to the quotient) — retained."""
    y_pos = torch.ones(H, W).cumsum(0).float().unsqueeze(0)
    x_pos = torch.ones(H, W).cumsum(1).float().unsqueeze(0)
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / d_model // 2))[:, None, None]
    pe = torch.zeros(d_model, H, W)
    pe[0::4] = torch.sin(x_pos * div)
    pe[1::4] = torch.cos(x_pos * div)
    pe[2::4] = torch.sin(y_pos * div)
    pe[3::4] = torch.cos(y_pos * div)
    return pe[None]


def linear_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """nndepth/blocks/attn_block.py:23-58 (no masks). q (N,L,H,D), k/v (N,S,H,D)."""
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    S = v.size(1)
    v = v / S
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S).contiguous()


def loftr_layer(sd: SD, p: str, x: torch.Tensor, source: torch.Tensor, nhead: int = 8) -> torch.Tensor:
    """nndepth/blocks/transformer.py:39-66."""
    N, L, Cm = x.shape
    D = Cm // nhead
    q = F.linear(x, sd[p + ".q_proj.weight"]).view(N, -1, nhead, D)
    k = F.linear(source, sd[p + ".k_proj.weight"]).view(N, -1, nhead, D)
    v = F.linear(source, sd[p + ".v_proj.weight"]).view(N, -1, nhead, D)
    m = linear_attention(q, k, v).view(N, -1, Cm)
    m = F.linear(m, sd[p + ".merge.weight"])
    m = F.layer_norm(m, (Cm,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    m = F.linear(torch.relu(F.linear(torch.cat([x, m], 2), sd[p + ".mlp.0.weight"])), sd[p + ".mlp.2.weight"])
    m = F.layer_norm(m, (Cm,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    return x + m


def feature_transformer(sd: SD, p: str, kind: str, f0: torch.Tensor, f1: torch.Tensor):
    """nndepth/blocks/transformer.py:98-121 with one layer: "self" or "cross" (note cross uses the UPDATED f0)."""
    lp = p + ".layers.0"
    if kind == "self":
        return loftr_layer(sd, lp, f0, f0), loftr_layer(sd, lp, f1, f1)
    f0 = loftr_layer(sd, lp, f0, f1)
    f1 = loftr_layer(sd, lp, f1, f0)
    return f0, f1


# ------------------------------------------------------------------------ instance-norm encoder (cre_stereo/model.py:70-72)
def _inorm(x: torch.Tensor) -> torch.Tensor:
    return F.instance_norm(x, eps=1e-5)


def _res_block_in(sd: SD, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    y = torch.relu(_inorm(_conv(sd, p + ".conv1", x, stride=stride, padding=1)))
    y = torch.relu(_inorm(_conv(sd, p + ".conv2", y, padding=1)))
    s = _inorm(_conv(sd, p + ".downsample.0", x, stride=stride))
    return torch.relu(s + y)


def basic_encoder_in(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """nndepth/encoders/basic_encoder.py:71-93 with norm_fn='instance' (InstanceNorm2d(affine=False): no parameters)."""
    x = torch.relu(_inorm(_conv(sd, p + ".conv1", x, stride=2, padding=3)))
    for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = _res_block_in(sd, f"{p}.{layer}.0", x, stride)
        x = _res_block_in(sd, f"{p}.{layer}.1", x, 1)
    return _conv(sd, p + ".conv2", x)


# ------------------------------------------------------------------------ a20: the cascade
def cre_stereo_forward(sd: SD, frame1: torch.Tensor, frame2: torch.Tensor, iters: int, hidden: int = 128,
                       flow_init: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """nndepth/models/cre_stereo/model.py:124-288 (fnet_ds = 8).  Returns the list of `up_disp` tensors
    (N,2,H,W): iters//2 (1/32 of the image... i.e. fmap/4) + iters//2 (fmap/2) + iters (fmap)."""
    ds = 8
    fm = basic_encoder_in(sd, "fnet", torch.cat([frame1, frame2], 0))
    fmap1, fmap2 = torch.split(fm, fm.shape[0] // 2, 0)
    f1_8, f2_8 = F.avg_pool2d(fmap1, 2, 2), F.avg_pool2d(fmap2, 2, 2)
    off8 = (torch.sigmoid(_conv(sd, "conv_offset_8", f1_8, padding=1)) - 0.5) * 2.0
    net, inp = torch.split(fmap1, [hidden, hidden], 1)
    net, inp = torch.tanh(net), torch.relu(inp)
    net8, inp8 = F.avg_pool2d(net, 2, 2), F.avg_pool2d(inp, 2, 2)
    f1_16, f2_16 = F.avg_pool2d(fmap1, 4, 4), F.avg_pool2d(fmap2, 4, 4)
    off16 = (torch.sigmoid(_conv(sd, "conv_offset_16", f1_16, padding=1)) - 0.5) * 2.0
    net16, inp16 = F.avg_pool2d(net, 4, 4), F.avg_pool2d(inp, 4, 4)
    N, Cf, H16, W16 = f1_16.shape
    pe = pos_enc_sine(256, frame1.shape[2] // (ds * 4), frame1.shape[3] // (ds * 4))[:, :, :H16, :W16]
    t1 = (f1_16 + pe).permute(0, 2, 3, 1).reshape(N, H16 * W16, Cf)
    t2 = (f2_16 + pe).permute(0, 2, 3, 1).reshape(N, H16 * W16, Cf)
    t1, t2 = feature_transformer(sd, "self_att_fn", "self", t1, t2)
    f1_16 = t1.reshape(N, H16, W16, Cf).permute(0, 3, 1, 2)
    f2_16 = t2.reshape(N, H16, W16, Cf).permute(0, 3, 1, 2)

    def cross(a, b):
        return feature_transformer(sd, "cross_att_fn", "cross", a, b)

    outs: List[torch.Tensor] = []

    def ub(n, i, c, fl):
        return update_block(sd, "update_block", n, i, c, fl)

    if flow_init is not None:
        scale = fmap1.shape[2] / flow_init.shape[2]
        flow = -scale * F.interpolate(flow_init, size=fmap1.shape[2:], mode="bilinear", align_corners=True)
    else:
        fl16 = torch.zeros(N, 2, H16, W16)
        up = None
        for it in range(iters // 2):
            corr = agcl_corr_att_offset(f1_16, f2_16, fl16, off16, it % 2 == 1, att=cross)
            net16, mask, delta = ub(net16, inp16, corr, fl16)
            fl16 = fl16 + delta
            up = convex_upsample(fl16, mask, ds)
            outs.append(up)
        scale = f1_8.shape[2] / up.shape[2]
        fl8 = scale * F.interpolate(up, size=f1_8.shape[2:], mode="bilinear", align_corners=True)
        for it in range(iters // 2):
            corr = agcl_corr_att_offset(f1_8, f2_8, fl8, off8, it % 2 == 1)
            net8, mask, delta = ub(net8, inp8, corr, fl8)
            fl8 = fl8 + delta
            up = convex_upsample(fl8, mask, ds)
            outs.append(up)
        scale = fmap1.shape[2] / up.shape[2]
        flow = scale * F.interpolate(up, size=fmap1.shape[2:], mode="bilinear", align_corners=True)
    for it in range(iters):
        corr = agcl_corr_iter(fmap1, fmap2, flow, it % 2 == 1)
        net, mask, delta = ub(net, inp, corr, flow)
        flow = flow + delta
        outs.append(convex_upsample(flow, mask, ds))
    return outs


def cre_two_stage_forward(sd: SD, frame1: torch.Tensor, frame2: torch.Tensor, iters: int) -> List[torch.Tensor]:
    """BASELINE.json config 5 harness: cascade on the half-resolution pair, then the full-resolution pair
    with flow_init = that result (the `flow_init` entry of cre_stereo/model.py:205-212, pinned by make_golden_cre.py)."""
    # half resolution, rounded up to a multiple of 32 (the cascade needs H/32 == (H/8)//4: model.py:172-174)
    h, w = -(-(frame1.shape[2] // 2) // 32) * 32, -(-(frame1.shape[3] // 2) // 32) * 32
    s1, s2 = (F.interpolate(f, size=(h, w), mode="bilinear", align_corners=True) for f in (frame1, frame2))
    coarse = cre_stereo_forward(sd, s1, s2, iters)
    return cre_stereo_forward(sd, frame1, frame2, iters, flow_init=coarse[-1])


# ------------------------------------------------------------------------ state-dict spec
def cre_stereo_spec(fnet_dim: int = 256, hidden: int = 128):
    """(key, shape) list of CREStereoBase.state_dict() (cre_stereo/model.py:70-101) in registration order."""
    spec = []

    def conv(name, co, ci, kh, kw):
        spec.append((name + ".weight", (co, ci, kh, kw)))
        spec.append((name + ".bias", (co,)))

    conv("fnet.conv1", 64, 3, 7, 7)
    cin = 64
    for layer, dim in (("layer1", 64), ("layer2", 96), ("layer3", 128)):
        for blk in (0, 1):
            p = f"fnet.{layer}.{blk}"
            conv(p + ".conv1", dim, cin, 3, 3)
            conv(p + ".conv2", dim, dim, 3, 3)
            conv(p + ".downsample.0", dim, cin, 1, 1)
            cin = dim
    conv("fnet.conv2", fnet_dim, 128, 1, 1)
    spec += update_block_spec("update_block", hidden, 36, hidden, 2, 8)
    for att in ("self_att_fn", "cross_att_fn"):
        p = att + ".layers.0"
        for lin in ("q_proj", "k_proj", "v_proj", "merge"):
            spec.append((f"{p}.{lin}.weight", (fnet_dim, fnet_dim)))
        spec.append((p + ".mlp.0.weight", (2 * fnet_dim, 2 * fnet_dim)))
        spec.append((p + ".mlp.2.weight", (fnet_dim, 2 * fnet_dim)))
        for nm in ("norm1", "norm2"):
            spec.append((f"{p}.{nm}.weight", (fnet_dim,)))
            spec.append((f"{p}.{nm}.bias", (fnet_dim,)))
    conv("conv_offset_16", 18, fnet_dim, 3, 3)
    conv("conv_offset_8", 18, fnet_dim, 3, 3)
    return spec
