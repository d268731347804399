"""Feature encoder used by RAFT-Stereo (reference: nndepth/encoders/basic_encoder.py:8-93,
nndepth/blocks/residual_block.py:6-60).

Parameter container for SURVEY.md §8f-1: module / parameter names match the reference so its checkpoints load unchanged.
At inference the model classes run the encoder as hand-written HIP (csrc/encoder.hip through ops.EncoderEngine, one C-ABI call:
nnd_encoder_forward) straight from these parameters; this module's own `forward` is the plain PyTorch-ROCm formulation and is
only reached through the explicit opt-out `hip_encoder=False` of the model classes (never silently).
"""
import torch
import torch.nn as nn


def _norm(kind: str, c: int) -> nn.Module:
    if kind == "batch":
        return nn.BatchNorm2d(c)
    if kind == "instance":
        return nn.InstanceNorm2d(c, affine=False)
    if kind == "group":
        return nn.GroupNorm(c // 8, c)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(f"norm_fn must be batch|group|instance|none, got {kind}")


class ResidualBlock(nn.Module):
    """Two 3x3 convs + a 1x1 projection shortcut that is applied unconditionally."""

    def __init__(self, in_planes: int, planes: int, norm_fn: str = "group", stride: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1, self.norm2, self.norm3 = (_norm(norm_fn, planes) for _ in range(3))
        # the shortcut's norm is the same object as norm3 -> appears under both names in state_dict
        self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)

    def forward(self, x):
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        return self.relu(self.downsample(x) + y)


class BasicEncoder(nn.Module):
    def __init__(self, output_dim: int = 128, norm_fn: str = "batch", dropout: float = 0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        dims, strides, cin = (64, 96, 128), (1, 2, 2), 64
        for i, (d, s) in enumerate(zip(dims, strides), start=1):
            setattr(self, f"layer{i}", nn.Sequential(ResidualBlock(cin, d, norm_fn, s), ResidualBlock(d, d, norm_fn, 1)))
            cin = d
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        self.dropout = nn.Dropout2d(dropout) if dropout > 0 else None

    def forward(self, x):
        pair = isinstance(x, (tuple, list))
        if pair:
            x = torch.cat(list(x), dim=0)
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        x = self.conv2(x)
        if self.dropout is not None:
            x = self.dropout(x)
        return torch.split(x, x.shape[0] // 2, dim=0) if pair else x
