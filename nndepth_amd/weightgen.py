"""Deterministic, torch-RNG-independent weight generator.

No trained checkpoint is available offline (reference `nndepth/models/README.md:13` is a
Drive link), so parity and benchmarks run on weights produced by a counter-based hash:
every tensor is a pure function of (state_dict key, shape).  The golden fixtures under
`tests/golden/` were produced by loading exactly these tensors into the reference model
(`oracle/make_golden.py`), and the GPU box regenerates the identical tensors.
"""
import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
GAIN = 0.8  # global scale on conv weights (chosen so the 32-iteration recurrence is well conditioned)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(tag: str, n: int) -> np.ndarray:
    """n float32 values in [0, 1), a pure function of (tag, index)."""
    seed = np.uint64(zlib.crc32(tag.encode("utf-8"))) << np.uint64(32)
    with np.errstate(over="ignore"):
        bits = _splitmix64(seed + np.arange(n, dtype=np.uint64))
    return ((bits >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def make_tensor(key: str, shape: Tuple[int, ...], dtype: torch.dtype = torch.float32) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    # ResidualBlock registers one norm module under two names (reference
    # nndepth/blocks/residual_block.py:51: `downsample = Sequential(conv, self.norm3)`), so both
    # state_dict keys must carry the same values.
    key = key.replace(".norm3.", ".downsample.1.")
    n = int(np.prod(shape)) if len(shape) else 1
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    u = uniform01(key, n)
    if key.endswith("running_var"):
        v = 0.8 + 0.4 * u
    elif key.endswith("running_mean"):
        v = 0.2 * u - 0.1
    elif len(shape) >= 2:  # conv / linear weight: U(-a, a), a = sqrt(3 / fan_in)
        fan_in = int(np.prod(shape[1:]))
        a = np.float32(GAIN * np.sqrt(3.0 / fan_in))
        v = (2.0 * u - 1.0) * a
    elif key.endswith("weight"):  # norm scale
        v = 0.8 + 0.4 * u
    else:  # bias
        v = 0.1 * u - 0.05
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).reshape(shape)).to(dtype)


def fill_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]]) -> Dict[str, torch.Tensor]:
    """spec: iterable of (key, shape) -> {key: tensor}."""
    return {k: make_tensor(k, s) for k, s in spec}


def fill_module_(module: torch.nn.Module, prefix: str = "") -> torch.nn.Module:
    """Overwrite every parameter/buffer of `module` in place with generated values."""
    sd = module.state_dict()
    new = {k: make_tensor(prefix + k, tuple(v.shape), v.dtype) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return module


def synthetic_frames(seed: int, batch: int, height: int, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Synthetic stereo pair in [-1, 1]: right = left shifted by a smooth disparity + noise,
    so the correlation volume has structure (not pure noise)."""
    n = batch * 3 * height * width
    base = uniform01(f"frame{seed}", n).reshape(batch, 3, height, width)
    # low-pass along x so neighbouring pixels correlate
    k = 9
    pad = np.pad(base, ((0, 0), (0, 0), (0, 0), (k, k)), mode="wrap")
    sm = sum(pad[..., i:i + width] for i in range(2 * k + 1)) / (2 * k + 1)
    left = (sm - sm.mean()) / (sm.std() + 1e-6)
    left = np.clip(left * 0.5, -1, 1).astype(np.float32)
    shift = 6 + (seed % 5)
    right = np.roll(left, -shift, axis=-1)
    noise = (uniform01(f"noise{seed}", n).reshape(left.shape) - 0.5) * 0.05
    right = np.clip(right + noise, -1, 1).astype(np.float32)
    return torch.from_numpy(left), torch.from_numpy(right)
