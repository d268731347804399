"""Feasibility study for SURVEY/VERDICT item 5 (a route past the 157 TFLOP/s fp32-MFMA ceiling): emulate on the CPU what a
split-bf16 MFMA conv computes — x = x0 + x1 + x2, w = w0 + w1 + w2 (bf16 pieces), products with i + j <= 2 accumulated in
fp32 — inside the oracle's RAFT-Stereo forward (update-block convs only), and measure the drift of the final up_disp against
the plain fp32 oracle on the TartanAir pair at iterations 1 / 4 / 12 / 32.  Dev-only; uses oracle/ as the checker.
    python scripts/study/split_bf16_numerics.py [nsplit_products]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import torch_ref as R  # noqa: E402
from nndepth_amd import weightgen  # noqa: E402


def split3(x, dtype, scale=1.0):
    """pieces of x*scale in `dtype`, returned divided by scale again (scale is a power of two: exact)"""
    if scale != 1.0:
        return tuple(p / scale for p in split3(x * scale, dtype))
    x0 = x.to(dtype).float()
    r1 = x - x0
    x1 = r1.to(dtype).float()
    r2 = r1 - x1
    x2 = r2.to(dtype).float()
    return x0, x1, x2


MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16x3_6"
PAIRS = {"bf16x3_6": (torch.bfloat16, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
         "bf16x3_3": (torch.bfloat16, [(0, 0), (0, 1), (1, 0)]),
         "f16x2_3": (torch.float16, [(0, 0), (0, 1), (1, 0)]),
         "f16x2_4": (torch.float16, [(0, 0), (0, 1), (1, 0), (1, 1)]),
         # range-scaled fp16 pieces: activations x 2^4, weights x 2^s with max|w| * 2^s in [2^13, 2^14) per layer, so that the low
         # pieces stay out of fp16's subnormals (the kernel undoes the power-of-two scales exactly in its epilogue)
         "f16x2_3s": (torch.float16, [(0, 0), (0, 1), (1, 0)])}
SCALED = MODE.endswith("s")
_wcache = {}
orig_conv = R._conv


def split_conv(sd, name, x, stride=1, padding=0):
    if not name.startswith("update_block.") or name.endswith("convf1") or name.endswith("flow_head.conv2"):
        return orig_conv(sd, name, x, stride, padding)  # VALU kernels stay fp32
    dtype, pairs = PAIRS[MODE]
    if name not in _wcache:
        w = sd[name + ".weight"]
        ws_ = 2.0 ** (13 - int(np.floor(np.log2(float(w.abs().max()))))) if SCALED else 1.0
        _wcache[name] = split3(w, dtype, ws_)
    ws, xs = _wcache[name], split3(x, dtype, 16.0 if SCALED else 1.0)
    # small terms first so they are not absorbed by the large partial sum (the kernel can order its MFMAs the same way)
    acc = None
    for (i, j) in sorted(pairs, key=lambda p: -(p[0] + p[1])):
        y = F.conv2d(xs[i], ws[j], None, stride=stride, padding=padding)
        acc = y if acc is None else acc + y
    return acc + sd[name + ".bias"].view(1, -1, 1, 1)


def main():
    torch.set_num_threads(8)
    from PIL import Image
    frames = []
    for side in ("left", "right"):
        img = np.asarray(Image.open(os.path.join(ROOT, "tests/golden", f"tartanair_000000_{side}.png")).convert("RGB"))
        t = torch.from_numpy(img.copy()).permute(2, 0, 1).float().unsqueeze(0)
        t = F.interpolate(t, (544, 960), mode="bilinear")
        frames.append((t - 127.5) / 127.5)
    sd = weightgen.fill_state_dict(R.raft_stereo_spec())
    g = dict(np.load(os.path.join(ROOT, "tests/golden/forward_tartanair.npz")))
    with torch.no_grad():
        t0 = time.time()
        ref, ref_low = R.raft_stereo_forward(sd, frames[0], frames[1], 32, return_lowres=True)
        print(f"fp32 oracle: {time.time() - t0:.1f} s; vs golden it32 {np.abs(ref[-1].numpy() - g['up_disp_it32']).max():.2e}")
        R._conv = split_conv
        t0 = time.time()
        got, got_low = R.raft_stereo_forward(sd, frames[0], frames[1], 32, return_lowres=True)
        R._conv = orig_conv
        print(f"{MODE}: {time.time() - t0:.1f} s")
    for it in (1, 4, 12, 32):
        e_up = (got[it - 1] - ref[it - 1]).abs().max().item()
        e_low = (got_low[it - 1] - ref_low[it - 1]).abs().max().item()
        print(f"  iter {it:2d}: up_disp max-abs vs fp32 oracle {e_up:.2e}   low-res {e_low:.2e}")
    print(f"  it32 vs reference golden: {np.abs(got[-1].numpy() - g['up_disp_it32']).max():.2e}")


if __name__ == "__main__":
    main()
