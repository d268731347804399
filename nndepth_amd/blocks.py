"""Drop-in for the reference's `BasicUpdateBlock` (nndepth/blocks/update_block.py:68-112).

An `nn.Module` with the same constructor arguments, the same `state_dict()` keys / shapes
(`encoder.convc1.weight` ... `mask.2.bias`, so `load_weights(..., strict=True)` keeps working)
and the same `forward(net, inp, corr, flow) -> (net, mask, delta_flow)`.  The parameters are
only containers: forward() repacks them once into MFMA fragment order and runs the HIP
update operator (csrc/update_block.hip); no torch conv is ever executed.
"""
from typing import Tuple, Union

import torch
import torch.nn as nn

from . import ops


def _conv(ci, co, k, pad):
    return nn.Conv2d(ci, co, k, padding=pad)


class BasicUpdateBlock(nn.Module):
    def __init__(self, hidden_dim: int, cor_planes: int, context_dim: int = 128, gru: str = "sep_conv",
                 flow_channel: int = 2, spatial_scale: Union[Tuple[int, int], int] = 8, arithmetic: str = "fp32"):
        super().__init__()
        sps = spatial_scale ** 2 if isinstance(spatial_scale, int) else spatial_scale[0] * spatial_scale[1]
        gin = hidden_dim + context_dim + hidden_dim
        enc = nn.Module()
        enc.convc1 = _conv(cor_planes, 256, 1, 0)
        enc.convc2 = _conv(256, 192, 3, 1)
        enc.convf1 = _conv(flow_channel, 128, 7, 3)
        enc.convf2 = _conv(128, 64, 3, 1)
        enc.conv = _conv(64 + 192, hidden_dim - flow_channel, 3, 1)
        self.encoder = enc
        g = nn.Module()
        if gru == "sep_conv":
            for n in ("convz1", "convr1", "convq1"):
                setattr(g, n, _conv(gin, hidden_dim, (1, 5), (0, 2)))
            for n in ("convz2", "convr2", "convq2"):
                setattr(g, n, _conv(gin, hidden_dim, (5, 1), (2, 0)))
        elif gru == "conv_gru":
            for n in ("convz1", "convr1", "convq1"):
                setattr(g, n, _conv(gin, hidden_dim, 3, 1))
        else:
            raise KeyError(gru)
        self.gru = g
        fh = nn.Module()
        fh.conv1 = _conv(hidden_dim, hidden_dim, 3, 1)
        fh.conv2 = _conv(hidden_dim, flow_channel, 3, 1)
        self.flow_head = fh
        self.mask = nn.Sequential(_conv(hidden_dim, hidden_dim * 2, 3, 1), nn.ReLU(inplace=True),
                                  _conv(hidden_dim * 2, sps * 9, 1, 0))
        # arithmetic of the fused loops' convolutions: "fp32" (exact fp32 MFMA, default) or "bf16x3" (csrc/conv_split.hip)
        self.engine = ops.UpdateBlockEngine(hidden_dim, context_dim, cor_planes, flow_channel, sps * 9, gru, arithmetic)
        self._packed_version = None

    def _version(self):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters())

    def sync_engine(self, device) -> ops.UpdateBlockEngine:
        """(Re)pack the parameters for the HIP kernels if they changed since the last call."""
        v = (self._version(), str(device))
        if v != self._packed_version:
            self.engine.load(self.state_dict(), device=device)
            self._packed_version = v
        return self.engine

    @torch.no_grad()
    def forward(self, net: torch.Tensor, inp: torch.Tensor, corr: torch.Tensor, flow: torch.Tensor):
        eng = self.sync_engine(net.device)
        return eng.forward(net.float(), inp.float(), corr.float(), flow.float())
