"""CREStereo with the MI355X-native hot path.

`CREStereoBase` keeps the reference's constructor kwargs, `state_dict()` keys (same registration order) and
`forward(frame1, frame2, flow_init=None) -> List[{"up_disp": (N,2,H,W)}]` of
nndepth/models/cre_stereo/model.py:17-288.  Inside `forward()`:

    encoder (instance norm)                              HIP  csrc/encoder.hip (nnd_encoder_forward, norm = instance)  model.py:139
    avg-pools, offset convs, sine position encoding, LoFTR self/cross attention
                              PyTorch-ROCm (adjacent rows)
    AGCL (warp + window correlation, offset sampling)     HIP  csrc/agcl.hip          model.py:204-206,229,252,277
    update block (2-channel flow)                         HIP  csrc/update_block.hip  model.py:231,254,279
    convex upsample (2-channel)                           HIP  csrc/corr1d.hip        model.py:234,257,282
  each stage of the cascade is ONE C-ABI call (`nnd_cre_stereo_refine`: all its iterations enqueued as a 3-stream
  DAG); `fused_loop=False` keeps the reference's seam-by-seam loop, every step still a HIP kernel.

The three stages of the cascade (1/32, 1/16 and 1/8 of the image for fnet_ds = 8) run `iters//2`, `iters//2` and
`iters` update iterations; even iterations search a 1x9 window, odd ones a 3x3 window.
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import NndError
from .blocks import BasicUpdateBlock
from .cost_volume import AGCL
from .encoder import BasicEncoder
from .raft_stereo import AutoCalibrate, hip_encoder_blocker, load_weights, require_eval
from .upsample import convex_upsample


# ------------------------------------------------------------------ adjacent PyTorch pieces (not the replaced path)
class LoFTREncoderLayer(nn.Module):
    """One LoFTR layer with linear attention (reference nndepth/blocks/transformer.py:8-66,
    nndepth/blocks/attn_block.py:16-58); parameter names match the reference."""

    def __init__(self, d_model: int, nhead: int):
        super().__init__()
        self.dim, self.nhead = d_model // nhead, nhead
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(2 * d_model, 2 * d_model, bias=False), nn.ReLU(),
                                 nn.Linear(2 * d_model, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, x: torch.Tensor, source: torch.Tensor) -> torch.Tensor:
        n = x.size(0)
        q = F.elu(self.q_proj(x).view(n, -1, self.nhead, self.dim)) + 1
        k = F.elu(self.k_proj(source).view(n, -1, self.nhead, self.dim)) + 1
        v = self.v_proj(source).view(n, -1, self.nhead, self.dim)
        s = v.size(1)
        kv = torch.einsum("nshd,nshv->nhdv", k, v / s)
        z = 1 / (torch.einsum("nlhd,nhd->nlh", q, k.sum(dim=1)) + 1e-6)
        msg = (torch.einsum("nlhd,nhdv,nlh->nlhv", q, kv, z) * s).reshape(n, -1, self.nhead * self.dim)
        msg = self.norm1(self.merge(msg))
        msg = self.norm2(self.mlp(torch.cat([x, msg], dim=2)))
        return x + msg


class LocalFeatureTransformer(nn.Module):
    """`forward(feat0, feat1)` takes (N, L, C) tokens like the reference (PyTorch path).  `forward_maps(map0, map1)` is the
    same computation on (N, C, H, W) maps — the tokens transposed — and runs in HIP (csrc/loftr.hip: every Linear is a 1x1
    conv of the map, the linear attention and the LayerNorms are small kernels); CREStereoBase and AGCL use it on the GPU."""

    def __init__(self, d_model: int, nhead: int, layer_names, attention: str = "linear"):
        super().__init__()
        if attention != "linear":
            raise ValueError("only the linear attention of CREStereo is provided")
        self.d_model, self.nhead, self.layer_names = d_model, nhead, list(layer_names)
        self.layers = nn.ModuleList([LoFTREncoderLayer(d_model, nhead) for _ in self.layer_names])
        self._engines, self._version = None, None

    def _layer_step(self, run, feat0, feat1):
        for i, name in enumerate(self.layer_names):
            if name == "self":
                feat0, feat1 = run(i, feat0, feat0), run(i, feat1, feat1)
            elif name == "cross":
                feat0 = run(i, feat0, feat1)
                feat1 = run(i, feat1, feat0)  # sees the updated feat0, like the reference
            else:
                raise KeyError(name)
        return feat0, feat1

    def forward(self, feat0: torch.Tensor, feat1: torch.Tensor):
        return self._layer_step(lambda i, a, b: self.layers[i](a, b), feat0, feat1)

    def hip_ready(self, t: torch.Tensor) -> bool:
        return t.is_cuda and not self.training and self.d_model // self.nhead == 32

    def forward_maps(self, map0: torch.Tensor, map1: torch.Tensor):
        if not self.hip_ready(map0):
            n, c, h, w = map0.shape
            a, b = self.forward(map0.permute(0, 2, 3, 1).reshape(n, h * w, c), map1.permute(0, 2, 3, 1).reshape(n, h * w, c))
            return a.reshape(n, h, w, c).permute(0, 3, 1, 2), b.reshape(n, h, w, c).permute(0, 3, 1, 2)
        v = (tuple((p.data_ptr(), p._version) for p in self.parameters()), str(map0.device))
        if v != self._version:
            self._engines = [ops.LoftrEngine(self.d_model, self.nhead).load(layer.state_dict(), device=map0.device)
                             for layer in self.layers]
            self._version = v
        return self._layer_step(lambda i, a, b: self._engines[i].forward(a.float(), b.float()), map0, map1)


# ------------------------------------------------------------------ the model
class CREStereoBase(AutoCalibrate, nn.Module):
    def __init__(self, fnet_cls: str = "basic_encoder", update_cls: str = "basic_update_block", iters: int = 12,
                 max_disp: int = 192, num_fnet_channels: int = 256, hidden_dim: int = 128, context_dim: int = 128,
                 search_num: int = 9, mixed_precision: bool = False, test_mode: bool = False, tracing: bool = False,
                 include_preprocessing: bool = False, weights: Optional[str] = None, strict_load: bool = True,
                 fused_loop: bool = True, hip_encoder: bool = True, arithmetic: str = "fp16x2", **kwargs):
        super().__init__()
        self.arithmetic = arithmetic  # update-block / encoder convolutions: "fp16x2" (default; 2 fp16 pieces, parity-gated), "bf16x3" (3 bf16 pieces) or "fp32" (exact fp32 MFMA)
        if fnet_cls != "basic_encoder" or update_cls != "basic_update_block":
            raise ValueError("CREStereoBase: only basic_encoder / basic_update_block exist (as in the reference)")
        if context_dim != hidden_dim:
            raise ValueError("Context dim must be equal to hidden_dim in this model")
        if search_num != 9:
            raise ValueError("AGCL searches 9 positions")
        self.max_flow, self.mixed_precision, self.test_mode, self.iters = max_disp, mixed_precision, test_mode, iters
        self.hidden_dim, self.context_dim, self.search_num = hidden_dim, context_dim, search_num
        self.tracing, self.include_preprocessing = tracing, include_preprocessing
        self.fused_loop = fused_loop
        self.hip_encoder = hip_encoder
        self._enc_engine, self._enc_version = None, None
        self.fnet = BasicEncoder(output_dim=num_fnet_channels, norm_fn="instance", dropout=0)
        self.fnet_ds = 8
        self.update_block = BasicUpdateBlock(hidden_dim=hidden_dim, cor_planes=4 * 9, flow_channel=2,
                                             context_dim=context_dim, spatial_scale=self.fnet_ds, arithmetic=arithmetic)
        self.self_att_fn = LocalFeatureTransformer(num_fnet_channels, 8, ["self"], "linear")
        self.cross_att_fn = LocalFeatureTransformer(num_fnet_channels, 8, ["cross"], "linear")
        self.conv_offset_16 = nn.Conv2d(num_fnet_channels, 2 * search_num, 3, padding=1)
        self.conv_offset_8 = nn.Conv2d(num_fnet_channels, 2 * search_num, 3, padding=1)
        self.range_16 = self.range_8 = 1
        self.corr_cls = AGCL  # seam: same call shape as the reference's AGCL
        self.weights, self.strict_load = weights, strict_load
        if weights is not None:
            load_weights(self, weights, strict_load)

    def convex_upsample(self, flow, mask, rate=4):
        return convex_upsample(flow, mask, rate)

    def forward_fnet(self, frame1: torch.Tensor, frame2: torch.Tensor):
        """fnet([frame1, frame2]) (model.py:139); on the GPU at inference the instance-norm encoder runs in HIP
        (csrc/encoder.hip, norm = instance: raw convs + per-sample statistics + one fused apply pass per block)."""
        if self.hip_encoder:
            why = hip_encoder_blocker(self.fnet, ("instance", "batch", "none"))
            if why:
                raise NndError(f"CREStereoBase: the HIP encoder cannot run this fnet ({why}); pass hip_encoder=False to run "
                               "the encoder's PyTorch-ROCm modules explicitly")
            tensors = list(self.fnet.state_dict().values())
            v = (tuple((t.data_ptr(), t._version) for t in tensors), str(frame1.device))
            if v != self._enc_version:
                if self._enc_engine is None:
                    self._enc_engine = ops.EncoderEngine(self.fnet.conv2.out_channels, self.fnet.norm_fn, 0, self.arithmetic)
                self._enc_engine.load(self.fnet.state_dict(), None, device=frame1.device)
                self._enc_version = v
            B = frame1.shape[0]
            fmaps, _ = self._enc_engine.forward(frame1.float(), frames_b=frame2.float())  # no torch.cat copy (basic_encoder.py:74-76)
            return fmaps[:B], fmaps[B:]
        return self.fnet([frame1, frame2])  # explicit opt-in (hip_encoder=False): PyTorch-ROCm modules

    def _offset_convs(self, device):
        """conv_offset_16 / conv_offset_8 packed for the MFMA conv (repacked when their parameters change)."""
        mods = (self.conv_offset_16, self.conv_offset_8)
        v = (tuple((p.data_ptr(), p._version) for m in mods for p in m.parameters()), str(device))
        if getattr(self, "_off_version", None) != v:
            self._off_engines = tuple(ops.Conv2d(m.weight, m.bias, device=device) for m in mods)
            self._off_version = v
        return self._off_engines

    def _stage(self, corr_fn, net, inp, flow, offset, n_iters: int, iter_mode: bool, outs: List[Dict[str, torch.Tensor]]):
        if self.fused_loop and isinstance(corr_fn, AGCL) and n_iters > 0:
            # ONE C-ABI call for the whole stage (nnd_cre_stereo_refine): AGCL + update block + advance + upsample
            eng = self.update_block.sync_engine(net.device)
            f1, f2 = corr_fn.attended()
            up, flow, net = eng.refine_cre(f1, f2, net.float(), inp.float(), self.fnet_ds, n_iters, flow_init=flow,
                                           extra_offset=None if iter_mode else offset.float())
            outs.extend({"up_disp": up[i]} for i in range(n_iters))
            return net, flow, up[n_iters - 1]
        up = None
        for itr in range(n_iters):
            corr = corr_fn(flow, offset, small_patch=(itr % 2 == 1), iter_mode=iter_mode)
            net, mask, delta = self.update_block(net, inp, corr, flow)
            flow = flow + delta
            up = self.convex_upsample(flow, mask, rate=self.fnet_ds)
            outs.append({"up_disp": up})
        return net, flow, up

    def forward(self, frame1: torch.Tensor, frame2: torch.Tensor, flow_init: Optional[torch.Tensor] = None,
                upsample: bool = True, test_mode: bool = False, **kwargs):
        require_eval(self)
        with torch.no_grad():
            return self._forward_calibrated(frame1, frame2, flow_init)

    def _forward(self, frame1: torch.Tensor, frame2: torch.Tensor, flow_init: Optional[torch.Tensor] = None):
        frame1, frame2 = frame1.contiguous(), frame2.contiguous()
        hd, ds = self.hidden_dim, self.fnet_ds
        fmap1, fmap2 = self.forward_fnet(frame1, frame2)
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        net, inp = ops.split_tanh_relu(fmap1, hd)  # split + tanh + relu in one kernel (model.py:148-151)
        outs: List[Dict[str, torch.Tensor]] = []
        if flow_init is not None:
            scale = fmap1.shape[2] / flow_init.shape[2]
            flow = ops.resize_bilinear_ac(flow_init.float(), fmap1.shape[2:], -scale)
        else:
            # both pooled scales of every map in one pass each (model.py:154-177)
            f1_8, f1_16 = ops.avg_pool_2x_4x(fmap1)
            f2_8, f2_16 = ops.avg_pool_2x_4x(fmap2)
            net8, net16 = ops.avg_pool_2x_4x(net)
            inp8, inp16 = ops.avg_pool_2x_4x(inp)
            conv16, conv8 = self._offset_convs(fmap1.device)
            # 1/(4*ds): attention-refined features, learned offsets, cross attention inside every AGCL call
            off16 = ops.conv2d_offset(conv16, f1_16, self.range_16)  # range * (sigmoid(conv) - 0.5) * 2 in the conv epilogue
            n, c, h16, w16 = f1_16.shape
            # x + PositionEncodingSine (pos_enc.py:22-42, model.py:180-196): one kernel for both maps, the table generated on the fly
            f1_pe, f2_pe = ops.pos_enc_sine_add(f1_16, f2_16)
            f1_16, f2_16 = self.self_att_fn.forward_maps(f1_pe, f2_pe)
            flow16 = torch.zeros(n, 2, h16, w16, dtype=torch.float32, device=fmap1.device)
            _, _, up = self._stage(self.corr_cls(f1_16, f2_16, att=self.cross_att_fn), net16, inp16, flow16, off16,
                                   self.iters // 2, False, outs)
            # 1/(2*ds): learned offsets, no attention
            off8 = ops.conv2d_offset(conv8, f1_8, self.range_8)
            scale = f1_8.shape[2] / up.shape[2]
            flow8 = ops.resize_bilinear_ac(up, f1_8.shape[2:], scale)  # scale * interpolate(bilinear, align_corners=True)
            _, _, up = self._stage(self.corr_cls(f1_8, f2_8), net8, inp8, flow8, off8, self.iters // 2, False, outs)
            scale = fmap1.shape[2] / up.shape[2]
            flow = ops.resize_bilinear_ac(up, fmap1.shape[2:], scale)
        # 1/ds: plain warped window correlation
        _, _, up = self._stage(self.corr_cls(fmap1, fmap2), net, inp, flow, None, self.iters, True, outs)
        if self.test_mode:
            return up
        return outs


def two_stage_forward(model: CREStereoBase, frame1: torch.Tensor, frame2: torch.Tensor) -> List[Dict[str, torch.Tensor]]:
    """The "2-stage cascaded" inference of BASELINE.json config 5 (the reference model has the `flow_init` hook,
    cre_stereo/model.py:205-212, but no caller): run the 3-scale cascade on the half-resolution pair, then the
    full-resolution pair with `flow_init` = the half-resolution result (one stage of `iters` iterations at 1/8)."""
    # half resolution, rounded up to a multiple of 32 (the cascade needs H/32 == (H/8)//4: model.py:172-174)
    h, w = -(-(frame1.shape[2] // 2) // 32) * 32, -(-(frame1.shape[3] // 2) // 32) * 32
    small = [ops.resize_bilinear_ac(f.float(), (h, w)) for f in (frame1, frame2)]  # interpolate(bilinear, align_corners=True)
    coarse = model(small[0], small[1])
    init = coarse if torch.is_tensor(coarse) else coarse[-1]["up_disp"]
    return model(frame1, frame2, flow_init=init)
